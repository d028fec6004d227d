# FETCH_SIZE / WRITE_SIZE per access width on this box (tools/ubench/fetch_calib.hip): bash tools/prof_calib.sh
# two counter passes (the two counters do not fit one pass), kernel trace only; -> gpurun_out/r04_fetch_calib.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ -x tools/ubench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
./tools/ubench/fetch_calib > gpurun_out/fetch_calib_bytes.json
for CTR in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/calib_$CTR && rocprofv3 --kernel-trace --pmc $CTR -d /tmp/calib_$CTR -o p -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > /dev/null 2> /tmp/calib_$CTR.log ) || tail -3 /tmp/calib_$CTR.log
done
python tools/calib_table.py gpurun_out/fetch_calib_bytes.json $(find /tmp/calib_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/calib_WRITE_SIZE -name "*.db" | head -1) > gpurun_out/r04_fetch_calib.txt
cat gpurun_out/r04_fetch_calib.txt
