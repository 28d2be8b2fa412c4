#!/bin/bash
# round 5, GPU call L: fm_search_fast2x_kernel pinned to five wavefronts per SIMD (amdgpu_waves_per_eu(5, 8)) against the default build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
for rep in 1 2; do
timeout 600 python tools/exp/time_fm_ilp.py 3000000000 10000000 100 2>&1 | grep -v amdgpu.ids | sed 's/^/default: /' | tee -a $O/summary.txt
BG_SO=tools/exp/_ko/libbiogpu_w5.so timeout 600 python tools/exp/time_fm_ilp.py 3000000000 10000000 100 2>&1 | grep -v amdgpu.ids | sed 's/^/w5:      /' | tee -a $O/summary.txt
done
