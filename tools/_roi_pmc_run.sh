#!/bin/bash
# r13: PMC passes over the ROIAlign launch on random ROIs, generation order vs dealt per XCD (profiles/r13_roialign_pmc.txt)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r13f
mkdir -p $OUT
db() { find $1 -name "*.db" | head -1; }
for mode in plain dealt; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $C | cut -d' ' -f1)
    rm -rf /tmp/p_roi
    rocprofv3 --kernel-trace --pmc $C -d /tmp/p_roi -o t -- python $REPO/tools/roi_pmc.py 0 1000 7 $mode > $OUT/roi_${mode}_$tag.log 2>&1
    echo "$mode --pmc $C" >> $OUT/roi_pmc.txt
    python $REPO/tools/rocpd_pmc.py $(db /tmp/p_roi) fpn_roi_align >> $OUT/roi_pmc.txt 2>&1
  done
done
cat $OUT/roi_pmc.txt
