cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/run42_err.txt
python tools/bench_index_build.py > gpurun_out/r02_index_build.json 2>/dev/null
python tools/bench_compare.py c4 > gpurun_out/r02_compare_c4.json 2>/dev/null
SMG_BENCH_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_forced_collectives.json 2>/dev/null
bash tools/prof_r02.sh compare > gpurun_out/run42_prof.txt 2>&1
bash tools/prof_r02.sh sketch >> gpurun_out/run42_prof.txt 2>&1
cat gpurun_out/r02_index_build.json; grep 'bitmatrix_kernel' gpurun_out/r02_compare_pmc.txt | cut -c1-140 | head -9
