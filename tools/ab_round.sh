#!/bin/bash
# A/B visit of the GPU box for kernel schedule variants (development aid). Usage: tools/ab_round.sh <tag>
TAG=${1:-ab}
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -x --tb=short -k "conv1x1 or winograd or deform or hipconv" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
(ONLY=1x1 NOTORCH=1 REPS=20 timeout 300 python tools/microbench_conv.py; ONLY=dcn NOTORCH=1 REPS=10 timeout 300 python tools/microbench_conv.py) > gpurun_out/${TAG}_micro.txt 2>&1
timeout 300 python tools/bench_winograd.py > gpurun_out/${TAG}_wino.txt 2>&1
B="timeout 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-configs2"
UPSNET_C1_KS=1 UPSNET_WINO_PF=0 UPSNET_DCN_VARIANT=1 $B > gpurun_out/${TAG}_bench_old.log 2>&1
UPSNET_C1_KS=2 UPSNET_WINO_PF=0 UPSNET_DCN_VARIANT=1 $B > gpurun_out/${TAG}_bench_c1.log 2>&1
UPSNET_C1_KS=1 UPSNET_WINO_PF=1 UPSNET_DCN_VARIANT=1 $B > gpurun_out/${TAG}_bench_pf.log 2>&1
UPSNET_C1_KS=1 UPSNET_WINO_PF=0 UPSNET_DCN_VARIANT=5 $B > gpurun_out/${TAG}_bench_dcn.log 2>&1
UPSNET_C1_KS=2 UPSNET_WINO_PF=1 UPSNET_DCN_VARIANT=5 $B > gpurun_out/${TAG}_bench_all.log 2>&1
UPSNET_C1_KS=2 UPSNET_WINO_PF=1 UPSNET_DCN_VARIANT=5 UPSNET_GRAPH_SLOTS=3 $B --in-flight 3 > gpurun_out/${TAG}_bench_all_if3.log 2>&1
for f in old c1 pf dcn all all_if3; do echo "$f: $(grep -o '"value": [0-9.]*' gpurun_out/${TAG}_bench_$f.log | head -1) $(grep -o '"ms_per_img_serial": [0-9.]*' gpurun_out/${TAG}_bench_$f.log | head -1)"; done | tee gpurun_out/${TAG}_summary.txt
