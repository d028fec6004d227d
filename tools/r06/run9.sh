cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
for b in 4 6 8 12 16; do echo "batches $b"; SMG_INGEST_BATCHES=$b timeout 900 python tools/bench_sketch_files.py 256 16 2>/dev/null; done
