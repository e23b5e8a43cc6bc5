#!/bin/bash
# Same-box A/B of two builds of libupsnet_hip.so (development aid): each twice, interleaved; prints value + serial ms + dense roofline.
# Usage (on the GPU box): tools/ab_lib.sh <tag> <prev.so> <new.so> [extra bench.py flags]
TAG=$1; A=$2; B=$3; shift 3
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out
i=0
for lib in $A $B $A $B; do
  i=$((i+1))
  env UPSNET_LIB_PATH=$lib timeout 600 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-configs2 "$@" > gpurun_out/${TAG}_ablib_$i.log 2>&1
  echo "$(basename $lib): $(tail -1 gpurun_out/${TAG}_ablib_$i.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('value %.2f serial %.3f ms | dense frac %.4f ms/img %.3f (wino %.3f direct %.3f) dcn %.4f ms' % (j['value'], j['ms_per_img_serial'], r['frac'], r['ms_per_image'], r['winograd']['ms_per_image'], r['direct']['ms_per_image'], r['deformable']['avg_launch_ms']))")"
done | tee gpurun_out/${TAG}_ablib.txt
