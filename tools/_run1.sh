cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -x -q -k "deconv or bf16 or cat or rpn" 2>&1 | tail -30 > gpurun_out/d2_test.log
timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/d2_bench_bf16.log 2>&1
tail -5 gpurun_out/d2_test.log; tail -1 gpurun_out/d2_bench_bf16.log | cut -c1-300
