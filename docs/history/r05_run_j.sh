#!/bin/bash
# round 5, GPU call J: index persistence (bg_fm_save / bg_fm_load) + the FM / pipeline suites
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fm.py tests/test_gpu_pipeline.py tests/test_gpu_fm_wide.py tests/test_gpu_sa_locate.py -m gpu -x -q 2>&1 | tail -25 | tee $O/summary.txt
