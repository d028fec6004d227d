#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/ubench/fetch_calib.hip) divided by the bytes each one is known to
move: the per-access-width ratios profiles/pmcfile.py applies before a `traffic` figure is quoted.

usage: calib_table.py <bytes.json printed by fetch_calib> <rocprofv3 db of the FETCH_SIZE pass> <db of the WRITE_SIZE pass>
output: a text table, one line per kernel:  calib  <kernel>  <counter>  <KiB per dispatch>  <known bytes>  <counted/known>  | shape"""
import json
import sqlite3
import sys


def counters(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    try:
        rows = cur.execute("select k.name, p.name, count(*), avg(e.value) from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
                           "join pmc_info p on p.id = e.pmc_id group by k.name, p.name").fetchall()
    except sqlite3.Error:
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    dur = {r[0]: r[1] for r in cur.execute("select name, avg(duration) from kernels group by name").fetchall()}
    return rows, dur


def main():
    known = json.load(open(sys.argv[1]))
    print("# FETCH_SIZE / WRITE_SIZE (KiB per dispatch, rocprofv3 --pmc, separate passes) of kernels that move a KNOWN number of bytes of a "
          "%d MiB buffer (8 x the Infinity Cache), by access shape; ratio = counted bytes / known bytes" % (known["buffer_bytes"] >> 20))
    print("# %-58s %-11s %16s %14s %8s %9s | shape" % ("kernel", "counter", "KiB/dispatch", "known bytes", "ratio", "GB/s"))
    for db in sys.argv[2:]:
        rows, dur = counters(db)
        for name, ctr, n, avg in sorted(rows):
            if "calib::" not in name:
                continue
            short = name.split("calib::")[1].split("(")[0]
            info = known["kernels"].get(short)
            if info is None:
                continue
            key = "read" if ctr == "FETCH_SIZE" else "write"
            if key not in info:
                continue
            counted = avg * 1024.0
            ns = dur.get(name, 0.0)
            print("calib  %-58s %-11s %16.1f %14d %8.3f %9.1f | %s" % (short, ctr, avg, info[key], counted / info[key],
                                                                      info[key] / ns if ns else 0.0, info["shape"]))


if __name__ == "__main__":
    main()
