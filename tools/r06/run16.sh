#!/bin/bash
# round 6: the 8-rank rehearsal of the bench launch on ONE GPU (every rank on device 0, RCCL refuses that, so the group is gloo + the
# device exchange paths); one parseable line of <= 8 KB from rank 0, transports and fell_back flags in it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
SMG_BENCH_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 2 --warmup 1 --bases 1e9 --no-io > $O/eight_ranks.json 2> $O/eight_ranks.err; echo "rc=$?" >> $O/eight_ranks.err
tail -3 $O/eight_ranks.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r06/eight_ranks.json") if x.startswith("{")]
print(len(l), "line(s)", [len(x) for x in l])
d=json.loads(l[-1]); print(d["value"], d["n_gpus"], d["config"]["comm"]["world_size_observed"], d["config"].get("collectives"))
s=d["summary"]; print({k:v for k,v in s.items() if k.startswith("dist_")})
PY
