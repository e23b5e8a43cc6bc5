set -x
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_bottleneck_bf16_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/bnk_test.log
timeout 600 python bench.py --conv-precision bf16 > gpurun_out/bnk_bench_bf16.log 2>&1
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bf16 -o bf16 -- python $R/bench.py --conv-precision bf16 --steps 20 --warmup 5 > $R/gpurun_out/prof_bf16.log 2>&1)
ls -R gpurun_out/prof_bf16 | head
tail -3 gpurun_out/bnk_test.log; tail -1 gpurun_out/bnk_bench_bf16.log | cut -c1-300
