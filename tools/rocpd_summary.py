#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases: per-kernel duration stats (kernel trace) or per-kernel
counter means (--pmc passes).  usage: rocpd_summary.py stats|pmc <results.db> [name-filter]"""
import sqlite3
import sys


def main():
    mode, db = sys.argv[1], sys.argv[2]
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    if mode == "stats":
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                         "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("name,calls,total_ns,avg_ns,min_ns,max_ns,percent")
        for n, k, s, a, lo, hi in rows:
            if filt in n:
                print(f'"{n}",{k},{s},{a:.0f},{lo},{hi},{100.0 * s / tot:.2f}')
    else:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        namec = "kernel_name" if "kernel_name" in cols else "name"
        rows = c.execute(f"select {namec}, counter_name, count(*), avg(value) from counters_collection "
                         f"group by {namec}, counter_name").fetchall()
        print("kernel,counter,dispatches,mean_value")
        for n, cn, k, v in rows:
            if filt in n:
                print(f'"{n}",{cn},{k},{v:.1f}')


if __name__ == "__main__":
    main()
