#!/bin/bash
# round 5, GPU call C: the 64-bit suffix-array / FM tests again (grid-stride kernels) and the 4.4 G-symbol FM run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fm_wide.py tests/test_gpu_sa_build.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -8 | tee -a $O/summary.txt
timeout 900 python tools/exp/fm_wide_big.py 4400000000 10000000 > $O/r05_fm_wide_4g4.json 2> $O/fm_wide.err; echo "fm_wide rc $?" | tee -a $O/summary.txt
cat $O/r05_fm_wide_4g4.json | tee -a $O/summary.txt; tail -8 $O/fm_wide.err
