cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/bench_gather.py > gpurun_out/r02_gather_c5.json 2>/dev/null
SMG_BENCH_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_forced_collectives.json 2>/dev/null
python tools/bench_compare.py c4 > gpurun_out/r02_compare_c4.json 2>/dev/null
cut -c150-400 gpurun_out/r02_gather_c5.json; cut -c1-300 gpurun_out/r02_compare_c4.json
