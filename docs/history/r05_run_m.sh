#!/bin/bash
# round 5, GPU call M: the 64-bit index's two-queries-per-quad kernel: forced-wide tests + persistence tests + the 4.4 G-symbol run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fm_wide.py tests/test_gpu_fm.py tests/test_gpu_sa_locate.py -m gpu -x -q 2>&1 | tail -4 | tee $O/summary.txt
timeout 900 python tools/exp/fm_wide_big.py 4400000000 10000000 > $O/r05_fm_wide_4g4.json 2> $O/fm_wide.err; echo "fm_wide rc $?" | tee -a $O/summary.txt
cat $O/r05_fm_wide_4g4.json | tee -a $O/summary.txt; tail -3 $O/fm_wide.err
