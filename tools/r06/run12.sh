cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
SMG_SIGLOAD_TRACE=1 timeout 900 python tools/bench_sigload.py 10000 > $O/sigload10k.json 2> $O/sigload10k.err; cat $O/sigload10k.json; grep sigload $O/sigload10k.err | tail -9
