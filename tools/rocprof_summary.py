#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite `*_results.db`, the default output of
`rocprofv3 --kernel-trace --stats` on ROCm 7.2) as a small text table for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--pmc] > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    print("# source: %s" % db)
    print("# kernel-trace summary (durations in microseconds)")
    print("%-110s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-110s %8d %14.3f %12.3f %8.2f" % (name[:110], calls, total, avg, pct))
    if "--pmc" in sys.argv:
        try:
            # pmc_events holds one row per (dispatch, counter, hardware instance); sum the
            # instances of a dispatch, then average over dispatches of the same kernel
            rows = list(c.execute(
                "select name, counter_name, count(*), avg(v), sum(v) from "
                "(select name, counter_name, dispatch_id, sum(counter_value) as v from pmc_events "
                " group by name, counter_name, dispatch_id) group by name, counter_name order by name"))
        except sqlite3.Error as ex:
            rows = []
            print("# pmc query failed: %s" % ex)
        if rows:
            print("\n# PMC counters per kernel (avg per dispatch)")
            print("%-90s %-18s %8s %16s" % ("kernel", "counter", "n", "avg"))
            for k, p, n, avg, _ in rows:
                print("%-90s %-18s %8d %16.1f" % (k[:90], p, n, avg))


if __name__ == "__main__":
    main()
