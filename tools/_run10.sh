cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_layerwise_gpu.py tests/test_ops_gpu.py tests/test_preprocess_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -8 > gpurun_out/r08h_pytest.log
tail -4 gpurun_out/r08h_pytest.log
bash tools/ab_lib.sh r08h $PWD/upsnet_amd/csrc/libupsnet_hip_prev.so $PWD/upsnet_amd/csrc/libupsnet_hip.so
