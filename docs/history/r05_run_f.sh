#!/bin/bash
# round 5, GPU call F: K3p in 512-thread blocks at 168 VGPRs (the chaining resident as a whole next to the fill) against round 4's launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
L=$R/rust-bio_amd/libbiogpu.so
timeout 300 python -m pytest tests/test_gpu_banded.py -m gpu -x -q 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 python tools/exp/time_banded_ab.py 65536 band_p_block256 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 300 python tools/exp/time_banded_ab.py 100000 band_p_block256 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 300 bash tools/exp/trace_banded_full.sh 65536 > $O/timeline_block512.txt 2>&1
BG_TRACE=1 timeout 200 python tools/exp/time_banded.py $L 65536 > $O/laps.log 2>&1
timeout 200 python tools/exp/time_banded.py $L 12500 2>&1 | grep -v amdgpu.ids | sed 's/^/12500 pairs: /' | tee -a $O/summary.txt
timeout 200 python tools/exp/time_banded.py $L 12500 band_p_block256=1 2>&1 | grep -v amdgpu.ids | sed 's/^/12500 pairs, block256: /' | tee -a $O/summary.txt
