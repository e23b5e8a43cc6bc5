cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_bottleneck_bf16_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/bnk_test.log
timeout 300 python tools/microbench_bottleneck.py > gpurun_out/bnk_micro.log 2>&1
timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/bnk_bench_bf16.log 2>&1
UPSNET_BF16_BLOCK_MIN_TILES=128 timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/bnk_bench_bf16_128.log 2>&1
tail -3 gpurun_out/bnk_test.log; cat gpurun_out/bnk_micro.log; tail -1 gpurun_out/bnk_bench_bf16.log | cut -c1-300; tail -1 gpurun_out/bnk_bench_bf16_128.log | cut -c1-300
