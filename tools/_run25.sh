cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/run25_pytest.txt 2>&1
python bench.py > gpurun_out/run25_bench.json 2> gpurun_out/run25_bench_err.txt
tail -4 gpurun_out/run25_pytest.txt; cut -c1-1500 gpurun_out/run25_bench.json; tail -3 gpurun_out/run25_bench_err.txt
