"""Timeline of the LAST image of a rocprofv3 (rocpd sqlite) kernel trace of bench.py: per dispatch its queue, start offset and
duration, plus the busy/idle picture of the device (union of the kernel intervals). Development aid.
Usage: python tools/rocpd_timeline.py <results.db> [anchor_kernel_substring] [n_images_back]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "image_to_nhwc4"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
starts = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(starts) < back + 1:
    sys.exit("anchor kernel %r seen %d times" % (anchor, len(starts)))
a, b = starts[-back - 1], starts[-back]
img = rows[a:b]
t0 = img[0][1]
print("image span %.3f ms, %d dispatches, queues: %s" % ((max(r[2] for r in img) - t0) / 1e6, len(img), sorted(set(r[3] for r in img))))
# busy union
ivs = sorted((r[1], r[2]) for r in img)
busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
gaps = []
for s, e in ivs[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e - t0, s - cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("device busy (union) %.3f ms; %d idle gaps totalling %.3f ms; largest: %s" % (
    busy / 1e6, len(gaps), sum(g[1] for g in gaps) / 1e6, ", ".join("%.0f us @%.2f ms" % (g[1] / 1e3, g[0] / 1e6) for g in sorted(gaps, key=lambda g: -g[1])[:8])))
for name, s, e, q in img:
    print("%8.3f ms  +%7.1f us  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, name[:100]))
