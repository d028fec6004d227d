cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gather.py tests/test_gpu_pair_ops.py tests/test_gpu_index_protocol.py -x -q 2>&1 | tail -5 > gpurun_out/run16_tests.txt
python tools/bench_gather.py > gpurun_out/run16_gather_default.json 2> gpurun_out/run16_err.txt
SMG_OVERLAP=ranges python tools/bench_gather.py > gpurun_out/run16_gather_ranges.json 2>> gpurun_out/run16_err.txt
bash tools/prof_gather.sh run16
cat gpurun_out/run16_tests.txt; cut -c1-600 gpurun_out/run16_gather_default.json; cut -c1-600 gpurun_out/run16_gather_ranges.json; cat gpurun_out/prof_run16.txt | cut -c1-150
