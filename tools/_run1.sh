cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -k "wreg or bf16" 2>&1 | tail -30 > gpurun_out/c1_test.log
timeout 300 python tools/microbench_conv1x1_bf16.py > gpurun_out/c1_micro.log 2>&1
timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/c1_bench_bf16.log 2>&1
tail -5 gpurun_out/c1_test.log; cat gpurun_out/c1_micro.log; tail -1 gpurun_out/c1_bench_bf16.log | cut -c1-300
