"""MFMA-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (rocpd sqlite):
util = MFMA busy cycles summed over the 1024 SIMDs / (active cycles x 1024). GRBM_GUI_ACTIVE is summed over the 8 XCDs.
Usage: python tools/mfma_util.py results.db [--total]
--total: also one line over ALL dispatches of the run -- with kernels of two images overlapping on the device (default bench: graph replay,
two in flight) the per-dispatch active cycles overlap in time, so the whole-run figure is busy cycles / (wall time of the kernels' union x
clock); it is printed as `sum of MFMA busy cycles` for tools/design_tables.py to divide by the bench's own wall time."""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
v = 'counters_collection' if 'counters_collection' in views else next(x for x in views if 'counter' in x and 'collection' in x)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % v)]
kn = 'kernel_name' if 'kernel_name' in cols else 'name'
cn = 'counter_name' if 'counter_name' in cols else next(c for c in cols if 'counter' in c and 'name' in c)
vn = 'value' if 'value' in cols else 'counter_value'
did = 'dispatch_id' if 'dispatch_id' in cols else cols[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for name, counter, value, d in cur.execute("select %s, %s, %s, %s from %s" % (kn, cn, vn, did, v)):
    acc[name][counter] += value
    nd[name].add(d)
rows = []
for name, c in acc.items():
    gui, busy = c.get('GRBM_GUI_ACTIVE', 0.0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    if busy > 0 and gui > 0:
        rows.append((gui, name, len(nd[name]), busy / (gui / 8.0 * 1024.0), gui / 8.0 / len(nd[name])))
print("| kernel | launches | avg active cycles / launch | MFMA pipe busy |")
print("|---|---|---|---|")
for gui, name, n, util, cyc in sorted(rows, reverse=True):
    print("| `%s` | %d | %.0f | %.1f %% |" % (name[:100], n, cyc, 100.0 * util))

if '--total' in sys.argv:
    tot_busy = sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for c in acc.values())
    tot_gui = sum(c.get('GRBM_GUI_ACTIVE', 0.0) for c in acc.values())
    nd_all = sum(len(v) for v in nd.values())
    print("\nwhole run: %d dispatches, sum of MFMA busy cycles %.0f (over 1024 SIMDs), sum of per-dispatch active cycles %.0f (8 XCDs)" % (nd_all, tot_busy, tot_gui))
    cols_k = [r[1] for r in cur.execute("pragma table_info(kernels)")] if 'kernels' in views else []
    if 'start' in cols_k and 'end' in cols_k:
        t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
        print("kernel span %.3f ms" % ((t1 - t0) / 1e6))
