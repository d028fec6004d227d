# one counter pass over a command on the GPU box: bash tools/prof_pmc.sh <tag> "<counters>" <command...>
# (counters in their own run with --kernel-trace only; summaries of the smg:: kernels go to gpurun_out/pmc_<tag>.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; CTR=$2; shift 2
( cd /tmp && rocprofv3 --kernel-trace --pmc $CTR -d /tmp/pmc_$TAG -o p -- "$@" > /dev/null 2> /tmp/pmc_$TAG.log ) || tail -3 /tmp/pmc_$TAG.log
python profiles/summarize.py $(find /tmp/pmc_$TAG -name "*.db" | head -1) | grep "smg::\|^kernel\|counter" > gpurun_out/pmc_$TAG.txt
