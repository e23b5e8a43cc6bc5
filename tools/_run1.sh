cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -k "wreg" 2>&1 | tail -3 > gpurun_out/w3_test.log
timeout 300 python tools/microbench_conv3x3_bf16.py > gpurun_out/w3_micro.log 2>&1
UPSNET_BNECK_256_8X8=1 timeout 300 python tools/microbench_bottleneck.py > gpurun_out/bnk_micro_8x8.log 2>&1
timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/w3_bench_bf16.log 2>&1
tail -2 gpurun_out/w3_test.log; cat gpurun_out/w3_micro.log; cat gpurun_out/bnk_micro_8x8.log; tail -1 gpurun_out/w3_bench_bf16.log | cut -c1-300
