cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_trunk_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_ref_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "nms or upsnet101 or proposals or mask_roi or parity or golden or reference" --tb=short 2>&1 | tail -30 > gpurun_out/r08c_pytest.log
tail -12 gpurun_out/r08c_pytest.log
cd /tmp && export TMPDIR=/tmp
for V in 0 2; do
UPSNET_ROI_KERNEL=$V UPSNET_OVERLAP=0 UPSNET_GRAPH=0 rocprofv3 --kernel-trace -d /tmp/p_trace$V -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-configs2 > $GRAFT_REPO_ROOT/gpurun_out/r08c_trace$V.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_calls.py $(find /tmp/p_trace$V -name "*.db" | head -1) fpn_roi_align nms_sort nms_mask nms_scan mask_removal prop_merge prop_sortk panoptic_fuse > $GRAFT_REPO_ROOT/gpurun_out/r08c_per_call_roi$V.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r08c_per_call_roi$V.txt
done
