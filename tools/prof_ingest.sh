# kernel + memory-copy timeline of the file ingest (GPU box).  Usage: bash tools/prof_ingest.sh
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/bench_ingest.py 4e9 2>&1 | tail -3
cat > /tmp/run_ingest.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from sourmash_amd.sketch import sketch_file
sketch_file("/tmp/synth.fa", "k=31,scaled=1000")
sketch_file("/tmp/synth.fa", "k=31,scaled=1000")
PY
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/prof_ingest -o ingest -- python /tmp/run_ingest.py > /tmp/prof_ingest.log 2>&1 || tail -5 /tmp/prof_ingest.log
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ingest -name "*.db" | head -1)
python profiles/summarize.py $DB
python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cp = cur.execute("select start, end, size from memory_copies where size >= 1048576 order by start").fetchall()
ks = cur.execute("select start, end from kernels order by start").fetchall()
busy = sum(e - s for s, e, _ in cp)
print(f"large copies: n={len(cp)} bytes={sum(c[2] for c in cp)} busy_ms={busy/1e6:.1f} GB/s_while_busy={sum(c[2] for c in cp)/busy:.1f}")
# overlap of copy intervals with kernel intervals
import bisect
starts = [k[0] for k in ks]
ov = 0
for s, e, _ in cp:
    i = max(bisect.bisect_left(starts, s) - 1, 0)
    while i < len(ks) and ks[i][0] < e:
        ov += max(0, min(e, ks[i][1]) - max(s, ks[i][0]))
        i += 1
print(f"copy time overlapped by kernels: {ov/1e6:.1f} ms of {busy/1e6:.1f} ms")
for s, e, n in cp[10:16]:
    print("copy", (s - cp[10][0]) / 1e3, (e - s) / 1e3, "us", n)
k0 = cp[10][0]
for s, e in [k for k in ks if cp[10][0] <= k[0] <= cp[16][0]][:24]:
    print("  kernel at", (s - k0) / 1e3, "dur", (e - s) / 1e3)
PY
