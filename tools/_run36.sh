cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_sp -o sp -- python $GRAFT_REPO_ROOT/tools/bench_compare_sparse.py > /tmp/prof_sp.log 2>&1 || tail -5 /tmp/prof_sp.log
cd $GRAFT_REPO_ROOT
python profiles/summarize.py $(find /tmp/prof_sp -name "*.db" | head -1) | grep -i "smg::\|^kernel\|rocprim" | head -30 | cut -c1-150
