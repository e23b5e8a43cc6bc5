cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_trunk_gpu.py -m gpu -q -p no:cacheprovider -k "roi_align or non_square or zero_fill or upsnet101" --tb=short 2>&1 | tail -30 > gpurun_out/r08b_pytest.log
tail -12 gpurun_out/r08b_pytest.log
timeout 600 python tools/microbench_roialign.py > gpurun_out/r08b_roialign.txt 2>&1
cat gpurun_out/r08b_roialign.txt
bash tools/profile_round.sh r08b
