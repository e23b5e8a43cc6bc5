cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_kernels_gpu.py tests/test_backward_gpu.py tests/test_golden_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -15
