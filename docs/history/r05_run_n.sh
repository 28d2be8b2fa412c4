#!/bin/bash
# round 5, GPU call N: non-temporal loads of the rank-block lines in fm_search_fast2x_kernel (variant build) against the default
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
for g in 3000000000; do
timeout 600 python tools/exp/time_fm_ilp.py $g 10000000 100 2>&1 | grep -v amdgpu.ids | sed 's/^/default: /' | tee -a $O/summary.txt
BG_SO=tools/exp/_ko/libbiogpu_nt.so timeout 600 python tools/exp/time_fm_ilp.py $g 10000000 100 2>&1 | grep -v amdgpu.ids | sed 's/^/nt:      /' | tee -a $O/summary.txt
done
