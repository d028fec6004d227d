cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --hip-trace --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/tools/probe_c3.py > /tmp/prof_c3.log 2>&1 || tail -5 /tmp/prof_c3.log
cd $GRAFT_REPO_ROOT
python - $(find /tmp/prof_c3 -name "*.db" | head -1) <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if "region" in t.lower() or "api" in t.lower()][:12])
for t in ("regions", "regions_and_samples"):
    if t in tabs:
        cols = [d[0] for d in cur.execute(f"select * from {t} limit 1").description]
        print(t, cols)
        rows = cur.execute(f"select name, count(*), sum(end-start), max(end-start) from {t} group by name order by sum(end-start) desc limit 18").fetchall()
        for r in rows: print(f"{r[0][:60]:<60} n={r[1]:>6} total_ms={r[2]/1e6:>9.3f} max_ms={r[3]/1e6:>8.3f}")
        break
PY
