#!/bin/bash
# round 5, GPU call E: K3p knock-outs (tools/exp/k3p_knockouts.sh) — the serial fill time of 32 768 pairs per variant
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
for v in "" _noring _noflush _nomerge _nohand _nochunk _w3; do
    L=$R/rust-bio_amd/libbiogpu.so
    [ -n "$v" ] && L=$R/tools/exp/_ko/libbiogpu$v.so
    timeout 200 python tools/exp/time_banded.py $L 32768 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
timeout 200 python tools/exp/time_banded.py $R/tools/exp/_ko/libbiogpu_w3.so 49152 chunk_pairs=24576 band_budget_gb=64 2>&1 | grep -v amdgpu.ids | sed 's/^/chunk 24576: /' | tee -a $O/summary.txt
timeout 200 python tools/exp/time_banded.py $R/rust-bio_amd/libbiogpu.so 49152 chunk_pairs=24576 band_budget_gb=64 2>&1 | grep -v amdgpu.ids | sed 's/^/chunk 24576: /' | tee -a $O/summary.txt
timeout 200 python tools/exp/time_banded.py $R/rust-bio_amd/libbiogpu.so 49152 2>&1 | grep -v amdgpu.ids | sed 's/^/chunk 16384: /' | tee -a $O/summary.txt
