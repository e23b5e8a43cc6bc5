REPO=/root/repo
OUT=$REPO/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$REPO
cd /tmp && export TMPDIR=/tmp
EAGER="env UPSNET_OVERLAP=0 UPSNET_GRAPH=0"
B="python $REPO/bench.py --no-cpu-baseline --no-configs2 --conv-precision bf16"
db() { find $1 -name "*.db" | head -1; }
$EAGER rocprofv3 --kernel-trace -d /tmp/p_trace -o t -- $B --steps 10 --warmup 5 > $OUT/bf16_trace_bench.log 2>&1
python $REPO/tools/rocpd_stats.py $(db /tmp/p_trace) 70 > $OUT/bf16_kernel_stats.txt 2>&1
python $REPO/tools/rocpd_timeline.py $(db /tmp/p_trace) > $OUT/bf16_timeline_serial.txt 2>&1
head -30 $OUT/bf16_kernel_stats.txt | cut -c1-160
