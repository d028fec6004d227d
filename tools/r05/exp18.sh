#!/bin/bash
# round 5, experiment 18: what the abundance join waits for -- counters of ap_join_kernel (separate --pmc passes, kernel trace only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp18; mkdir -p $O
S=$GRAFT_REPO_ROOT/profiles/summarize.py
pass() {
  local tag=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/p18/$tag -o p -- "$@" > /dev/null 2> /tmp/p18_$tag.log ) || tail -3 /tmp/p18_$tag.log
}
db() { find /tmp/p18/$1 -name "*.db" | head -1; }
B="python $GRAFT_REPO_ROOT/tools/bench_compare_ext.py"
pass A SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $B
pass B SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -- $B
pass C SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA -- $B
python $S $(db A) $(db B) $(db C) > $O/join_pmc.txt
grep -i "ap_join" $O/join_pmc.txt | cut -c1-160
