#!/bin/bash
# One GPU-box visit: parity tests + calibration + bench + rocprof summary. Usage: tools/gpu_round.sh <tag>
TAG=${1:-r01}
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
(rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; ls /root/reference 2>&1 | head -2) > gpurun_out/${TAG}_box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x --tb=short > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -40 gpurun_out/${TAG}_pytest.log
