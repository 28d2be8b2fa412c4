#!/bin/bash
# round 5, GPU call Q: fm_search_fast2x_kernel with a 32-bit query index (84 VGPRs) and the same compiled for six wavefronts per SIMD (80 VGPRs, 0-2 spills)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
for v in "" q32 q32w6; do
  so=""; [ -n "$v" ] && so=tools/exp/_ko/libbiogpu_$v.so
  BG_SO=$so timeout 600 python tools/exp/time_fm_ilp.py 3000000000 10000000 100 2>&1 | grep -v amdgpu.ids | sed "s/^/[$v] /" | tee -a $O/summary.txt
done
BG_SO=tools/exp/_ko/libbiogpu_q32w6.so timeout 600 python tools/exp/time_fm_ilp.py 100000000 10000000 100 2>&1 | grep -v amdgpu.ids | sed "s/^/[q32w6 100M] /" | tee -a $O/summary.txt
