cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_index_protocol.py tests/test_gpu_counter_protocol.py -x -q 2>&1 | tail -5 > gpurun_out/run19_tests.txt
python tools/bench_gather.py > gpurun_out/run19_a.json 2> gpurun_out/run19_err.txt
bash tools/prof_gather.sh run19
cat gpurun_out/run19_tests.txt; cut -c1-330 gpurun_out/run19_a.json; echo; cut -c1-330 gpurun_out/run19_split.json; echo; grep "build_range\|stream_lookup" gpurun_out/prof_run19.txt gpurun_out/prof_run19s.txt | cut -c1-150
