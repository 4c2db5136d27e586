"""Summarise the kernel-dispatch table of a rocprofv3 rocpd database (``*_results.db``) as the same table
``rocprofv3 --kernel-trace --stats`` prints: name, calls, total / average / min / max duration (ns), share.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for name, n, tot, avg, mn, mx in rows:
        print(f"\"{name}\",{n},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
