for lib in "" sourmash_amd/libsourmash_amd_r16.so; do
  echo "== library ${lib:-default (BR_RANGE 32768)}"
  for i in 1 2; do SMG_LIBRARY=$lib timeout 200 python tools/bench_gather.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather build', d['index_build_ms'], 'loop', d['loop_ms'], 'checks', all(d['checks'].values()))"; done
done
SMG_LIBRARY=sourmash_amd/libsourmash_amd_r16.so timeout 400 python -m pytest tests/test_gpu_gather.py tests/test_gpu_parallel.py -x -q 2>&1 | tail -2
bash tools/prof_r02.sh gather 2>&1 | grep -i "range_kernel\|scatter" | cut -c1-150
