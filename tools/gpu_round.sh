#!/bin/bash
# One GPU-box visit: parity tests + bench. Usage: tools/gpu_round.sh <tag> [pytest args...]
TAG=${1:-r08}; shift
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
(rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > gpurun_out/${TAG}_box.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --tb=short --durations=15 "$@" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -60 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --steps 40 --warmup 6 > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1500
