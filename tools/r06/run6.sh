cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
SMG_INGEST_TRACE=1 timeout 900 python tools/bench_sketch_files.py 256 16 > $O/many.json 2> $O/many.err; cat $O/many.json; grep "batch of" $O/many.err | tail -30
