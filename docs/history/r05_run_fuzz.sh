#!/bin/bash
# round 5: differential fuzz against the CPU oracle on the round's final sources
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05fuzz
mkdir -p $O
cd $R
echo "# csrc $(python tools/csrc_hash.py)" | tee $O/summary.txt
run() { local name=$1; shift; timeout 400 python "$@" > $O/$name.log 2>&1; echo "$* : $(grep -v amdgpu.ids $O/$name.log | tail -1)" | tee -a $O/summary.txt; }
run fm tests/fuzz_fm.py 20260927 240
run banded_k3p tests/fuzz_banded.py 20260927 200 k3p
run banded tests/fuzz_banded.py 20260928 150
run banded_long tests/fuzz_banded_long.py 20260927 200 12000
run pairwise tests/fuzz_pairwise.py 27 150
run pk16 tests/fuzz_pk16.py 28 150
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tee -a $O/summary.txt
