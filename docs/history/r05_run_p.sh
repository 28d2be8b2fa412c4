#!/bin/bash
# round 5, GPU call P: resident blocks per CU of fm_search_fast2x_kernel (5 by its registers) limited to 4 / 3 / 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
for b in 5 4 3 2; do
BG_FM_BLOCKS_PER_CU=$b timeout 600 python tools/exp/time_fm_ilp.py 3000000000 10000000 100 2>&1 | grep -v amdgpu.ids | sed "s/^/blocks $b: /" | tee -a $O/summary.txt
done
BG_FM_BLOCKS_PER_CU=4 timeout 600 python tools/exp/time_fm_ilp.py 100000000 10000000 100 2>&1 | grep -v amdgpu.ids | sed "s/^/100M blocks 4: /" | tee -a $O/summary.txt
