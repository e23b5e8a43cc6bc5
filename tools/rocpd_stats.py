"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table rocprofv3 --stats prints.
Usage: python tools/rocpd_stats.py <results.db> [top_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for name, calls, tot, avg, mn, mx in rows[:top]:
    short = name if len(name) < 110 else name[:107] + "..."
    print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" % (short, calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
print("\ntotal kernel time: %.3f ms over %d dispatches, %d distinct kernels" % (total / 1e6, sum(r[1] for r in rows), len(rows)))
