# per-kernel time of the gather index build and loop at C5 (GPU box); $1 = tag of the output files
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-gather}
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o gather -- python $GRAFT_REPO_ROOT/tools/bench_gather.py > /tmp/prof_$TAG.log 2>&1 || tail -5 /tmp/prof_$TAG.log
cd $GRAFT_REPO_ROOT
grep "^{" /tmp/prof_$TAG.log | tail -1 | cut -c1-400 > gpurun_out/prof_$TAG.txt
python profiles/summarize.py $(find /tmp/prof_$TAG -name "*.db" | head -1) | grep -i "smg::\|kernel" | head -24 >> gpurun_out/prof_$TAG.txt
