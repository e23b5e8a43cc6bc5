"""Per-call durations of selected kernels from a rocprofv3 (rocpd sqlite) kernel trace: python tools/rocpd_calls.py results.db substr [substr...]
Prints, per kernel name containing one of the substrings, the sorted list of per-call durations (us) -- e.g. the box-head and
mask-head ROIAlign calls or the three batched NMS calls of an image, which a per-kernel average hides."""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
subs = sys.argv[2:]
calls = collections.defaultdict(list)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcol = next((c for c in ('grid_x', 'grid_size_x', 'grid_size', 'workgroup_size_x') if c in cols), None)
ocol = 'start' if 'start' in cols else cols[0]
for name, dur, grid in db.execute("select name, duration, %s from kernels order by %s" % (gcol or "0", ocol)):
    if any(s in name for s in subs):
        calls[(name.split('(')[0], grid)].append(dur / 1e3)
for (name, grid), d in sorted(calls.items()):
    d = sorted(d)
    print("%-44s grid_x %-7s calls %3d | min %.1f p50 %.1f max %.1f us" % (name[:44], grid, len(d), d[0], d[len(d) // 2], d[-1]))
