#!/bin/bash
# round 5, GPU call K: quad-parallel seed fetch of fm_search_fast2x_kernel<SEEDS>: parity suites + the seed-and-extend leg on 3 Gbp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fm.py tests/test_gpu_pack2.py -m gpu -x -q 2>&1 | tail -5 | tee $O/summary.txt
timeout 900 python bench.py --skip-cpu --skip-fm --skip-k1 --skip-banded --skip-ingest --skip-packed --skip-semiglobal --pairs 65536 --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a gpurun_out/r05k/summary.txt
import json
d = json.loads([l for l in open("gpurun_out/r05k/bench.json") if l.startswith("{")][-1])
se = d.get("seed_extend", {})
print("seed_extend", se.get("value"), se.get("ms_per_step"), se.get("kernel_ms"))
print("fm_big", d.get("fm_big", {}).get("value"))
PY
