# per-kernel time of the gather loop at C5 (GPU box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_gather -o gather -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /tmp/prof_gather.log 2>&1 || tail -5 /tmp/prof_gather.log
cd $GRAFT_REPO_ROOT
grep "^{" /tmp/prof_gather.log | tail -1 | cut -c1-400
python profiles/summarize.py $(find /tmp/prof_gather -name "*.db" | head -1) | grep -i "smg::\|kernel" | head -20
