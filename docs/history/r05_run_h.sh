#!/bin/bash
# round 5, GPU call H: K5 with two queries per quad (fm option ilp2): parity tests, A/B at 100 Mbp and 3 Gbp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fm.py tests/test_gpu_pack2.py -m gpu -x -q 2>&1 | tail -4 | tee -a $O/summary.txt
timeout 400 python tools/exp/time_fm_ilp.py 100000000 10000000 100 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 600 python tools/exp/time_fm_ilp.py 3000000000 10000000 100 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
