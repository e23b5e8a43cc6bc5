#!/bin/bash
# One GPU-box visit producing everything profiles/<tag>_* and the generated tables of DESIGN.md are written from. Usage: tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace of the eager, single-stream bench (a kernel's duration is its own): kernel stats, per-call durations, timeline
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs, kernel-trace only -- never with other trace domains)
#   3. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE: per kernel (serial) and over the default run (graph replay, two images in flight)
#   4. per-launch table of the convolution kernels (algorithmic TFLOP/s and GB/s from the bench's own events)
#   5. the default bench line (graph replay, overlapped streams, cpu_baseline, configs2) + the other workloads + the bf16 mode's own line
#   6. the parity summary: worst error / bound of the strict fp64 tests at full size
#   7. bf16 mode (configs[2]): eager serial kernel stats + timeline, micro-benchmarks of its kernels, its bench line on configs[1] and configs[3]
TAG=${1:-r08}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$REPO
cd /tmp && export TMPDIR=/tmp
# 0. the default bench line on the fresh box (what the driver's BENCH run measures): before anything else has heated the chip
(cd $REPO && python bench.py > $OUT/${TAG}_bench.log 2>&1)
EAGER="env UPSNET_OVERLAP=0 UPSNET_GRAPH=0"
B="python $REPO/bench.py --no-cpu-baseline --no-configs2 --no-wide-offsets --no-roialign"
db() { find $1 -name "*.db" | head -1; }
$EAGER rocprofv3 --kernel-trace -d /tmp/p_trace -o t -- $B --steps 10 --warmup 5 > $OUT/${TAG}_trace_bench.log 2>&1
python $REPO/tools/rocpd_stats.py $(db /tmp/p_trace) 70 > $OUT/${TAG}_kernel_stats.txt 2>&1
python $REPO/tools/rocpd_calls.py $(db /tmp/p_trace) fpn_roi_align nms_sort nms_mask nms_scan dcn_fused panoptic_fuse mask_removal prop_merge prop_sortk mroi_finalize > $OUT/${TAG}_per_call.txt 2>&1
python $REPO/tools/rocpd_timeline.py $(db /tmp/p_trace) > $OUT/${TAG}_timeline_serial.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  $EAGER rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o t -- $B --steps 3 --warmup 3 > $OUT/${TAG}_pmc_$C.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(db /tmp/p_$C) > $OUT/${TAG}_pmc_$C.txt 2>&1
done
$EAGER rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o t -- $B --steps 3 --warmup 3 > $OUT/${TAG}_pmc_mfma.log 2>&1
python $REPO/tools/mfma_util.py $(db /tmp/p_mfma) > $OUT/${TAG}_mfma_util_serial.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma2 -o t -- $B --steps 20 --warmup 6 > $OUT/${TAG}_pmc_mfma_graph.log 2>&1
python $REPO/tools/mfma_util.py $(db /tmp/p_mfma2) --total > $OUT/${TAG}_mfma_util_graph.txt 2>&1
# bf16 mode (BASELINE configs[2]): eager serial trace -> kernel stats + timeline; the micro-benchmarks of its kernels against the paths they replace
B16="python $REPO/bench.py --no-cpu-baseline --no-configs2 --no-wide-offsets --no-roialign --conv-precision bf16"
$EAGER rocprofv3 --kernel-trace -d /tmp/p_trace16 -o t -- $B16 --steps 10 --warmup 5 > $OUT/${TAG}_bf16_trace_bench.log 2>&1
python $REPO/tools/rocpd_stats.py $(db /tmp/p_trace16) 45 > $OUT/${TAG}_bf16_kernel_stats.txt 2>&1
python $REPO/tools/rocpd_timeline.py $(db /tmp/p_trace16) > $OUT/${TAG}_bf16_timeline_serial.txt 2>&1
cd $REPO
(python tools/microbench_bottleneck.py; python tools/microbench_conv3x3_bf16.py; python tools/microbench_conv1x1_bf16.py) > $OUT/${TAG}_bf16_micro.txt 2>&1
python tools/microbench_stem.py > $OUT/${TAG}_stem_micro.txt 2>&1
python tools/bench_winograd36.py > $OUT/${TAG}_winograd36_micro.txt 2>&1
(hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/ubench/mfma_valu.hip && /tmp/mfma_valu) > $OUT/${TAG}_mfma_valu.txt 2>&1
# r13: the small-tile kernels against what they replace, the split-K F(4x4) form, the ROI -> XCD dealing
python tools/bench_conv1x1_ksw.py > $OUT/${TAG}_conv1x1_ksw.txt 2>&1
python tools/bench_conv3x3_ksw.py > $OUT/${TAG}_conv3x3_ksw.txt 2>&1
python tools/bench_winograd36_splitk.py > $OUT/${TAG}_winograd36_splitk.txt 2>&1
python tools/bench_roi_xcd_order.py > $OUT/${TAG}_roi_xcd_order.txt 2>&1
UPSNET_WINO36=0 python bench.py --no-cpu-baseline --no-configs2 --no-wide-offsets --no-roialign > $OUT/${TAG}_bench_ab_no_wino36.log 2>&1
python tools/make_pmc_json.py $TAG $OUT/${TAG}_pmc_FETCH_SIZE.txt $OUT/${TAG}_pmc_WRITE_SIZE.txt > $OUT/${TAG}_conv_pmc.json 2> $OUT/${TAG}_conv_pmc.err
cp $OUT/${TAG}_conv_pmc.json profiles/${TAG}_conv_pmc.json   # (so that the bench line of this very run can report roofline.traffic)
python tools/layer_table.py > $OUT/${TAG}_layer_table.txt 2>&1
python tools/microbench_roialign.py > $OUT/${TAG}_roialign.txt 2>&1
# (the headline line is taken FIRST, on the box as the driver finds it -- see below; here the same command again after ~3 minutes of sustained
# load: the chip's clock has settled lower, every kernel of the roofline sample is 5-9 % slower)
python bench.py --no-cpu-baseline --no-configs2 --no-wide-offsets > $OUT/${TAG}_bench_hot.log 2>&1
python bench.py --steps 150 --warmup 8 --no-cpu-baseline --no-configs2 --in-flight 1 > $OUT/${TAG}_bench_serial.log 2>&1
python bench.py --steps 400 --warmup 8 --no-cpu-baseline --no-configs2 --conv-precision bf16 > $OUT/${TAG}_bench_bf16.log 2>&1
python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-configs2 --conv-precision bf16x3 > $OUT/${TAG}_bench_bf16x3.log 2>&1
python bench.py --steps 250 --warmup 8 --no-cpu-baseline --no-configs2 --conv-precision bf16 --workload upsnet101dcn_coco_800x1333 > $OUT/${TAG}_bench_c3_bf16.log 2>&1
python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-configs2 --workload upsnet101dcn_mixed_1024x2048_800x1333 > $OUT/${TAG}_bench_c4.log 2>&1
# r11: the serial window in the PRODUCTION mode (one HIP graph, two streams), one image in flight: which kernels overlap, which wait
cd /tmp
rocprofv3 --kernel-trace -d /tmp/p_tl_graph -o t -- python $REPO/bench.py --no-cpu-baseline --no-configs2 --no-wide-offsets --no-roialign --in-flight 1 --steps 24 --warmup 8 > $OUT/${TAG}_trace_graph.log 2>&1
python $REPO/tools/rocpd_timeline.py $(db /tmp/p_tl_graph) image_to_nhwc4 6 > $OUT/${TAG}_timeline_graph_serial.txt 2>&1
# r11: FETCH / WRITE counters of configs[3]'s own workload (bench.py reports roofline.traffic only from a profile of the SAME workload and precision)
for C in FETCH_SIZE WRITE_SIZE; do
  $EAGER rocprofv3 --kernel-trace --pmc $C -d /tmp/p_c3_$C -o t -- $B --workload upsnet101dcn_coco_800x1333 --steps 3 --warmup 3 > $OUT/${TAG}_c3_pmc_$C.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(db /tmp/p_c3_$C) > $OUT/${TAG}_c3_pmc_$C.txt 2>&1
done
cd $REPO
python tools/make_pmc_json.py ${TAG}_c3 $OUT/${TAG}_c3_pmc_FETCH_SIZE.txt $OUT/${TAG}_c3_pmc_WRITE_SIZE.txt upsnet101dcn_coco_800x1333 fp32 > $OUT/${TAG}_c3_conv_pmc.json 2> $OUT/${TAG}_c3_conv_pmc.err
cp $OUT/${TAG}_c3_conv_pmc.json profiles/${TAG}_c3_conv_pmc.json
python tools/layer_table.py c3 > $OUT/${TAG}_layer_table_c3.txt 2>&1
python bench.py --steps 130 --warmup 8 --no-configs2 --no-cpu-baseline --workload upsnet101dcn_coco_800x1333 > $OUT/${TAG}_bench_c3.log 2>&1   # (again, now that its own PMC profile exists: roofline.traffic of THIS workload)
python bench.py --gpus 1 --dry-run > $OUT/${TAG}_dry_run.log 2>&1
timeout 900 python -m pytest tests/test_trunk_gpu.py tests/test_layerwise_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "worst|launches|passed|failed" > $OUT/${TAG}_parity.txt
tail -1 $OUT/${TAG}_bench.log | cut -c1-300
head -12 $OUT/${TAG}_kernel_stats.txt
