# merge-kernel variants (SMG_CMP_VARIANT): timing + identity with the bit-row path.  GPU box.
cd $GRAFT_REPO_ROOT
for v in 0 1 2 10 11 12; do
echo "== variant $v"
SMG_CMP_VARIANT=$v python - <<'PY' 2>&1 | grep -v amdgpu
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_compare as b
b.run(1000, True)
b.run(4000, False)
b.run(1000, False, pool=5_000_000, keep=1000)
b.run(1000, False, pool=5000, keep=10)
PY
done
