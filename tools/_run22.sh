cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_index_protocol.py tests/test_gpu_counter_protocol.py tests/test_gpu_parallel.py -x -q 2>&1 | tail -5 > gpurun_out/run22_tests.txt
SMG_GATHER_TRACE=1 python tools/bench_gather.py > gpurun_out/run22_a.json 2> gpurun_out/run22_trace.txt
bash tools/prof_r02.sh gather > gpurun_out/run22_prof.txt 2>&1
cat gpurun_out/run22_tests.txt; cut -c1-400 gpurun_out/run22_a.json; echo; tail -4 gpurun_out/run22_trace.txt; cat gpurun_out/run22_prof.txt | cut -c1-170
