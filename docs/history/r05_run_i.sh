#!/bin/bash
# round 5, GPU call I: the whole GPU suite, fuzz_fm on the two-queries-per-quad kernel (60 s), the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -8 | tee -a $O/summary.txt
timeout 200 python tests/fuzz_fm.py 20260926 60 > $O/fuzz_fm.log 2>&1; tail -3 $O/fuzz_fm.log | tee -a $O/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
tail -c 600 $O/bench.json
