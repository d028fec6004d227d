#!/bin/bash
# round 5: counters of the run-time-k sketch kernel (sketch_dna_words_kernel) at k = 128 and k = 200, the unrolled kernel at k = 88 beside them
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp33; mkdir -p $O
S=$GRAFT_REPO_ROOT/profiles/summarize.py
pass() {
  local tag=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/p33/$tag -o p -- "$@" > /dev/null 2> /tmp/p33_$tag.log ) || tail -3 /tmp/p33_$tag.log
}
db() { find /tmp/p33/$1 -name "*.db" | head -1; }
for k in 88 128 200; do
  B="python $GRAFT_REPO_ROOT/tools/bench_sketch_one_k.py $k 5e8 3"
  pass A$k SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- $B
  pass B$k SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -- $B
  { echo "== k = $k"; python $S $(db A$k) $(db B$k) | grep -i "sketch_dna"; } >> $O/long_k_pmc.txt
done
cat $O/long_k_pmc.txt | cut -c1-150
