"""Per-kernel averages of PMC counters from a rocprofv3 (rocpd sqlite) database. Usage: python tools/rocpd_pmc.py results.db [substr]"""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ''
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
    if sub and sub not in name:
        continue
    acc[name][counter] += value
    nd[name].add(d)
for name in sorted(acc, key=lambda n: -sum(acc[n].values())):
    n = len(nd[name])
    print("%s | dispatches %d | %s" % (name[:110], n, ", ".join("%s=%.1f/launch" % (c, s / n) for c, s in sorted(acc[name].items()))))
