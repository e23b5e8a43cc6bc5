"""Launches per image by origin from a rocprofv3 kernel trace (rocpd sqlite): python tools/launch_census.py results.db images
Groups: ours (libupsnet_hip.so kernels), hipBLASLt (Cijk_*), ATen (at::native / softmax), runtime (copyBuffer / fillBuffer)."""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
n_img = float(sys.argv[2])
grp = collections.defaultdict(lambda: [0, 0.0])
names = collections.defaultdict(lambda: [0, 0.0])
for name, cnt, tot in db.execute("select name, count(*), sum(duration) from kernels group by name"):
    if name.startswith('Cijk'):
        g = 'hipBLASLt'
    elif 'at::native' in name or 'softmax' in name or 'at::' in name:
        g = 'ATen'
    elif 'rocclr' in name:
        g = 'runtime copy/fill'
    else:
        g = 'ours'
    grp[g][0] += cnt
    grp[g][1] += tot
    if g in ('ATen', 'runtime copy/fill'):
        names[name[:90]][0] += cnt
        names[name[:90]][1] += tot
for g, (c, t) in sorted(grp.items()):
    print("%-18s %7.2f launches/img %9.1f us/img" % (g, c / n_img, t / 1e3 / n_img))
for nme, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][0]):
    print("   %6.2f/img %7.1f us/img  %s" % (c / n_img, t / 1e3 / n_img, nme))
