cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do python tools/bench_gather.py 2>/dev/null | cut -c150-330; done
for i in 1 2; do SMG_BENCH_GC=1 python tools/bench_gather.py 2>/dev/null | cut -c150-330; done
