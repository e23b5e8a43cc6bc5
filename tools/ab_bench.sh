#!/bin/bash
# Same-box A/B of the end-to-end bench over one environment knob (development aid): each value twice, interleaved.
# Usage (on the GPU box): tools/ab_bench.sh <tag> <ENV_VAR> <value A> <value B> [extra bench.py flags]
# e.g. tools/ab_bench.sh r07 UPSNET_CONV1X1_PAIR 0 1        tools/ab_bench.sh r07 UPSNET_DCN_VARIANT 5 6 --workload upsnet101dcn_coco_800x1333
TAG=$1; VAR=$2; A=$3; B=$4; shift 4
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for v in $A $B $A $B; do
  env $VAR=$v timeout 600 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-configs2 "$@" > gpurun_out/${TAG}_ab_${VAR}_$v.log 2>&1
  echo "$VAR=$v: $(grep -o '"value": [0-9.]*' gpurun_out/${TAG}_ab_${VAR}_$v.log | head -1) $(grep -o '"ms_per_img_serial": [0-9.]*' gpurun_out/${TAG}_ab_${VAR}_$v.log | head -1)"
done | tee gpurun_out/${TAG}_ab_${VAR}.txt
