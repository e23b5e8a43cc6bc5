#!/bin/bash
# r13: FETCH / WRITE / L2 hit counters of the F(4x4) kernel alone at four (Cin, Cout) (profiles/r13_wino36_traffic.txt)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r13h
mkdir -p $OUT
db() { find $1 -name "*.db" | head -1; }
for shape in "256 256" "256 64" "256 128" "64 256"; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/p_w36
    rocprofv3 --kernel-trace --pmc $C -d /tmp/p_w36 -o t -- python $REPO/tools/wino36_pmc.py $shape > $OUT/w36.log 2>&1
    echo "Cin Cout = $shape --pmc $C" >> $OUT/wino36_pmc.txt
    python $REPO/tools/rocpd_pmc.py $(db /tmp/p_w36) conv_wino36 >> $OUT/wino36_pmc.txt 2>&1
  done
done
cat $OUT/wino36_pmc.txt
