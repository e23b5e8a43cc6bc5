cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_ref_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "nms or proposals or mask_roi or parity or golden or reference or panoptic or mask_removal" --tb=short 2>&1 | tail -30 > gpurun_out/r08e_pytest.log
tail -5 gpurun_out/r08e_pytest.log
cd /tmp && export TMPDIR=/tmp
UPSNET_OVERLAP=0 UPSNET_GRAPH=0 rocprofv3 --kernel-trace -d /tmp/p_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-configs2 > $GRAFT_REPO_ROOT/gpurun_out/r08e_trace.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_calls.py $(find /tmp/p_trace -name "*.db" | head -1) fpn_roi_align nms_sort nms_mask nms_scan mask_removal mask_bits prop_ mroi_ panoptic_fuse pan_tail > $GRAFT_REPO_ROOT/gpurun_out/r08e_per_call.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r08e_per_call.txt
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find /tmp/p_trace -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r08e_timeline.txt 2>&1
