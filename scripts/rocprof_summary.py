#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) result into the small text summary kept under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof1/dist_results.db > profiles/r01_x.md
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main(path, title=""):
    c = sqlite3.connect(path)
    print("# rocprofv3 --kernel-trace --stats summary%s" % (": " + title if title else ""))
    print("\nsource: `%s`\n" % path)
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---|---|---|---|")
    for name, calls, total, avg, pct in c.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (short(name), calls, total, avg, pct))
    try:
        rows = list(c.execute("select name, value from pmc_events limit 1"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n## counters (sum over dispatches, per kernel)\n")
        q = ("select k.name, p.name, sum(p.value), count(*) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id group by k.name, p.name")
        try:
            print("| kernel | counter | sum | dispatches |\n|---|---|---|---|")
            for kn, pn, v, n in c.execute(q):
                print("| `%s` | %s | %.6g | %d |" % (short(kn, 60), pn, v, n))
        except sqlite3.Error as e:
            print("(counter join failed: %s)" % e)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
