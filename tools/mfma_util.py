"""MFMA-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (rocpd sqlite):
util = MFMA busy cycles summed over the 1024 SIMDs / (active cycles x 1024). GRBM_GUI_ACTIVE is summed over the 8 XCDs.
Usage: python tools/mfma_util.py results.db"""
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
