cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stem_pool_bf16_gpu.py tests/test_model_gpu.py -x -q -k "stem or bf16" 2>&1 | tail -30 > gpurun_out/sp_test.log
timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/sp_bench_bf16.log 2>&1
UPSNET_BF16_STEM=0 timeout 600 python bench.py --conv-precision bf16 --no-cpu-baseline --no-configs2 > gpurun_out/sp_bench_bf16_off.log 2>&1
tail -12 gpurun_out/sp_test.log; tail -1 gpurun_out/sp_bench_bf16.log | cut -c1-300;  tail -1 gpurun_out/sp_bench_bf16_off.log | cut -c1-300
