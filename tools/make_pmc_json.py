"""Turn the two PMC passes of tools/profile_round.sh (rocpd_pmc.py output for FETCH_SIZE and WRITE_SIZE) into profiles/<tag>_conv_pmc.json.
Usage: python tools/make_pmc_json.py <tag> <fetch.txt> <write.txt> [workload [precision]] > profiles/<tag>[_<workload>]_conv_pmc.json
Counters are KiB per launch at the L2 <-> fabric boundary (they include Infinity-Cache hits). Correction per
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE is doubled for 16 B/lane streaming reads (gfx950 counts
their 128-B requests at 64 B) -- and, measured here in r11 with tools/ubench/fetch_calib.hip (profiles/r11_fetch_calibration.txt), for EVERY
load width and for the NHWC gather shapes alike: FETCH_SIZE = 0.5000 x bytes read, WRITE_SIZE = 1.0000 x bytes written; WRITE_SIZE is used as reported. The json carries the source hash of the build it was measured on;
bench.py reports `roofline.traffic` only from a file whose hash matches the running build."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upsnet_amd import build as B

tag, fpath, wpath = sys.argv[1:4]
workload = sys.argv[4] if len(sys.argv) > 4 else 'upsnet50_cityscapes_1024x2048'
precision = sys.argv[5] if len(sys.argv) > 5 else 'fp32'


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?) \| dispatches (\d+) \| (.*)", line.strip())
        if not m:
            continue
        vals = dict(re.findall(r"(\w+)=([\d.]+)/launch", m.group(3)))
        if counter in vals:
            out[m.group(1).strip()] = (int(m.group(2)), float(vals[counter]))
    return out


fetch, write = parse(fpath, 'FETCH_SIZE'), parse(wpath, 'WRITE_SIZE')
# the launches bench.py's roofline counts as the dense family (upsnet_amd/ops.py, PROFILE events of kind 'conv'): the three convolution
# kernels, the block-boundary pair kernels and the fused stem; a split-K launch's reduce kernel belongs to that launch (bytes, no count)
DENSE = ('conv_igemm_f32_kernel', 'conv_wino16_f32_kernel', 'conv_wino36_f32_kernel', 'conv_wino16_tail_f32_kernel', 'conv1x1_frag_f32_kernel', 'conv1x1_pair_f32_kernel',
         'conv1x1_pair32_f32_kernel', 'stem_pool_f32_kernel', 'conv1x1_ksw_f32_kernel', 'conv3x3_ksw_f32_kernel')
EXTRA = ('conv_splitk_reduce',)
per, n_tot, f_tot, w_tot = {}, 0, 0.0, 0.0
for name, (n, f) in fetch.items():
    w = write.get(name, (n, 0.0))[1]
    if any(k in name for k in DENSE + EXTRA + ('dcn_fused_f32_kernel', 'fpn_roi_align', 'panoptic_fuse')):
        per[name] = dict(launches=n, fetch_kib=round(f, 1), write_kib=round(w, 1), hbm_bytes_per_launch=int((2 * f + w) * 1024))
    if any(k in name for k in DENSE + EXTRA):
        n_tot += 0 if any(k in name for k in EXTRA) else n
        f_tot += f * n
        w_tot += w * n
doc = {
    'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `UPSNET_OVERLAP=0 UPSNET_GRAPH=0 python '
              'bench.py --steps 3 --warmup 3 --no-cpu-baseline` (tools/profile_round.sh %s), 1x MI355X; per-kernel averages via tools/rocpd_pmc.py' % tag,
    'correction': 'hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: FETCH doubled for 16 B/lane streaming reads (gfx950 counts 128-B requests '
                  'at 64 B, MI355X_MICROARCH.md); counters sit at the L2 <-> fabric boundary and include Infinity-Cache hits',
    'srchash': B._source_hash(), 'workload': workload, 'precision': precision,
    'kernel': 'dense convolution family (' + ' + '.join(DENSE) + '; split-K reduce passes added to their launches), %d launches' % n_tot,
    'fetch_kb_per_launch_raw': round(f_tot / max(n_tot, 1), 1), 'write_kb_per_launch_raw': round(w_tot / max(n_tot, 1), 1),
    'hbm_bytes_per_launch': int((2 * f_tot + w_tot) / max(n_tot, 1) * 1024),
    'per_kernel': per,
}
print(json.dumps(doc, indent=1))
