#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (gpurun_out/prof_*/..._results.db) into the small text
summaries committed under profiles/.   usage: summarize.py <db> [<db> ...] > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|reduce_by_key_impl_wrapped_config|"
                  r"reduce_by_key_init_kernel|partition_kernel|select)", name)
    if "rocprim" in name and m:
        return "rocprim::" + m.group(1)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<smg::OwGeom<(\d+), [^>]*> >", r"<OwGeom\1>", name)       # overlap_lean_kernel<smg::OwGeom<25, 8192, ...> > -> <OwGeom25>
    # overlap_lean_kernel<smg::OwGeom<25, 10240, ...>, 2> -> overlap_lean_kernel<2> (0: overlaps, 1: builder's pass 1 counting, 2: staging)
    name = re.sub(r"<smg::OwGeom<(\d+), [^>]*>, (\d+)>", r"<\2>", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print(f"== {path}")
        # (min/max are shown because one kernel name can cover launches of different sizes, e.g. the
        #  1e9-byte CPU-baseline sample next to the 1e10-byte bench steps)
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                           "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                           "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':<62} {'calls':>5} {'total_ms':>10} {'avg_us':>11} {'min_us':>10} {'max_us':>10} {'%':>6} vgpr sgpr lds grid wg")
        shown = rows[:14] + [r for r in rows[14:] if "smg::" in r[0]]        # the top of the list, and every kernel of this library
        for r in shown:
            print(f"{short(r[0]):<62} {r[1]:>5} {r[2]/1e6:>10.3f} {r[3]/1e3:>11.2f} {r[4]/1e3:>10.2f} {r[5]/1e3:>10.2f} "
                  f"{100*r[2]/tot:>6.2f} {r[6]} {r[7]} {r[8]} {r[9]} {r[10]}")
        try:
            pm = cur.execute("select k.name, p.name, count(*), avg(e.value), sum(e.value) from pmc_events e "
                             "join kernels k on k.dispatch_id = e.dispatch_id join pmc_info p on p.id = e.pmc_id "
                             "group by k.name, p.name order by k.name").fetchall()
        except sqlite3.Error:
            pm = []
        if not pm:
            try:
                cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
                pm2 = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from "
                                  "counters_collection group by kernel_name, counter_name").fetchall() \
                    if "kernel_name" in cols else []
                pm = pm2
            except sqlite3.Error as e:
                print("  (no counters:", e, ")")
        if pm:
            print(f"  {'kernel':<50} {'counter':<22} {'dispatches':>10} {'avg/dispatch':>16} {'sum':>18}")
            for k, c, n, avg, tot_ in pm:
                if "smg::" in k:
                    print(f"  {short(k):<50} {c:<22} {n:>10} {avg:>16.1f} {tot_:>18.1f}")
        print()


if __name__ == "__main__":
    main()
