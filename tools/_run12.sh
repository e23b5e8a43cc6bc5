cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp
UPSNET_OVERLAP=0 UPSNET_GRAPH=0 rocprofv3 --kernel-trace -d /tmp/p_bf16 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-configs2 --conv-precision bf16 > $GRAFT_REPO_ROOT/gpurun_out/r08i_trace_bf16.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p_bf16 -name "*.db" | head -1) 40 > $GRAFT_REPO_ROOT/gpurun_out/r08i_kernel_stats_bf16.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find /tmp/p_bf16 -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r08i_timeline_bf16.txt 2>&1
head -45 $GRAFT_REPO_ROOT/gpurun_out/r08i_kernel_stats_bf16.txt | cut -c1-140
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r08i_trace_bf16.log | cut -c1-300
