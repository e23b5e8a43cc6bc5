"""profiles/<tag>_kernel_stats.md from the files tools/profile_round.sh leaves in gpurun_out/ (development aid).
Usage: python tools/make_stats_md.py <tag> <file with the 'what changed in this build' paragraph>"""
import os
import sys

tag, note = sys.argv[1], open(sys.argv[2]).read().strip()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
rd = lambda n: open(os.path.join(out, '%s_%s' % (tag, n))).read().rstrip()
stats = rd('kernel_stats.txt')
print("# %s — rocprofv3 --kernel-trace of `UPSNET_OVERLAP=0 UPSNET_GRAPH=0 python bench.py --steps 10 --warmup 5 --no-cpu-baseline` "
      "(1x MI355X, UPSNet-50 1024x2048, fp32)\n" % tag)
print(note + "\n")
print("Eager + single stream, so a kernel's duration is its own (the default bench replays one HIP graph per image on two overlapping "
      "streams). 40 images in the trace (per-image = calls / 40). Tools: tools/profile_round.sh %s -> tools/rocpd_stats.py.\n" % tag)
print(stats + "\n")
print("Per-call durations of the kernels a per-kernel average hides (tools/rocpd_calls.py, same trace):\n\n```\n" + rd('per_call.txt') + "\n```\n")
print("bench.py line of the same build and box (default flags: HIP graph replay, overlapped streams, two images in flight, cpu_baseline 1 + 3 "
      "passes;\n`roofline.traffic` comes from profiles/%s_conv_pmc.json, whose source hash equals this build's):\n\n```\n" % tag
      + rd('bench.log').split('\n')[-1] + "\n```")
