#!/bin/bash
# One GPU-box visit producing what profiles/<tag>_* is written from. Usage: tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace of the eager, single-stream bench (per-kernel averages not inflated by co-running kernels)
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs, kernel-trace only -- never with other trace domains)
#   3. the default bench line (graph replay, overlapped streams, cpu_baseline included)
TAG=${1:-r05}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EAGER="env UPSNET_OVERLAP=0 UPSNET_GRAPH=0"
$EAGER rocprofv3 --kernel-trace -d /tmp/p_trace -o t -- python $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-configs2 > $OUT/${TAG}_trace_bench.log 2>&1
python $REPO/tools/rocpd_stats.py $(find /tmp/p_trace -name "*.db" | head -1) 70 > $OUT/${TAG}_kernel_stats.txt 2>&1
python $REPO/tools/rocpd_calls.py $(find /tmp/p_trace -name "*.db" | head -1) fpn_roi_align nms_sort nms_mask nms_scan dcn_fused panoptic_fuse > $OUT/${TAG}_per_call.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  $EAGER rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o t -- python $REPO/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-configs2 > $OUT/${TAG}_pmc_$C.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(find /tmp/p_$C -name "*.db" | head -1) > $OUT/${TAG}_pmc_$C.txt 2>&1
done
cd $REPO
python tools/make_pmc_json.py $TAG $OUT/${TAG}_pmc_FETCH_SIZE.txt $OUT/${TAG}_pmc_WRITE_SIZE.txt > $OUT/${TAG}_conv_pmc.json 2> $OUT/${TAG}_conv_pmc.err
cp $OUT/${TAG}_conv_pmc.json profiles/${TAG}_conv_pmc.json   # (so that the bench line of this very run can report roofline.traffic)
python bench.py > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log | cut -c1-300
head -12 $OUT/${TAG}_kernel_stats.txt
