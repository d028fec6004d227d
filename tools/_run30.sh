cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_compare.py c4 > /tmp/prof_c4.log 2>&1 || tail -5 /tmp/prof_c4.log
cd $GRAFT_REPO_ROOT
python profiles/summarize.py $(find /tmp/prof_c4 -name "*.db" | head -1) | grep -i "smg::\|^kernel" | head -30 > gpurun_out/run30_prof.txt
cat gpurun_out/run30_prof.txt | cut -c1-150
