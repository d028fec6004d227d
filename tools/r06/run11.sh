cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 1500 python tools/bench_sigload.py 10000 100000 > $O/sigload.json 2> $O/sigload.err; echo rc=$?; tail -3 $O/sigload.err; cat $O/sigload.json
