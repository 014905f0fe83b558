#!/usr/bin/env python3
"""rocprofv3 results database (-d DIR -o NAME => DIR/NAME_results.db) -> markdown summary: per-kernel call count, total and
average duration (`--kernel-trace --stats` view), and, when the run collected PMC counters, their per-kernel sums."""
import sqlite3
import sys


def main(path, title):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# %s\n" % title)
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name.split("(")[0]
        print("| %s | %d | %.3f | %.1f | %.2f |" % (short, calls, total / 1e3, avg, pct))
    try:
        rows = list(cur.execute("select name, counter_name, sum(counter_value), count(*) from pmc_events group by name, "
                                "counter_name order by 3 desc"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n| kernel | counter | sum over dispatches | dispatches |\n|---|---|---|---|")
        for k, c, v, n in rows:
            print("| %s | %s | %.6g | %d |" % (k.split("(")[0], c, v, n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
