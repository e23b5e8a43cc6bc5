cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_stem_pool_gpu.py tests/test_layerwise_gpu.py tests/test_trunk_gpu.py tests/test_model_gpu.py tests/test_no_library_conv_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/spf_test.log
timeout 600 python bench.py --no-cpu-baseline --no-configs2 --steps 60 --warmup 8 > gpurun_out/spf_bench.log 2>&1
UPSNET_STEM_POOL=0 timeout 600 python bench.py --no-cpu-baseline --no-configs2 --steps 60 --warmup 8 > gpurun_out/spf_bench_off.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-configs2 --steps 60 --warmup 8 > gpurun_out/spf_bench2.log 2>&1
tail -12 gpurun_out/spf_test.log; for f in spf_bench spf_bench_off spf_bench2; do tail -1 gpurun_out/$f.log | cut -c1-260; done
