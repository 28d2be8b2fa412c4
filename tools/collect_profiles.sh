#!/bin/bash
# Collect the profiles/ artefacts of one round on the GPU box (run through gpurun):
#   tools/collect_profiles.sh <tag>        e.g. r02
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (all legs)   -> <tag>_bench_kernel_stats.csv
# 2. separate PMC passes, kernel-trace only (never combined with other trace domains):
#      FETCH_SIZE, WRITE_SIZE                                                      -> <tag>_pmc_traffic.json
#      SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES                   -> <tag>_sq_counters.json
#      SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS      -> <tag>_lds_counters.json (+ .txt table)
# 3. a calibration of FETCH/WRITE_SIZE against kernels with known traffic          -> <tag>_pmc_calibration.txt
# Everything lands in gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --skip-cpu"
# the counter passes keep one launch shape per kernel: the seed-and-extend leg reuses K2 / K5 on other batch sizes
PMCBENCH="$BENCH --skip-fmd --skip-pipeline --skip-packed --fm-big-genome 0 --banded-pairs 98304 --pipeline-reads-total 0 --steps 2 --warmup 0"
# ... and the FM kernel on the 3 Gbp index gets passes of its own (same kernel name as the 100 Mbp leg)
# ... together with the seed-and-extend leg on the same index (its kernels have names of their own: the SEEDS flavour of K5,
# K6, the se_* stages, the semiglobal flavour of K1p; K2 is shared with the 65 536-pair headline leg of this command)
BIGBENCH="$BENCH --skip-fmd --skip-fm --skip-k1 --skip-banded --skip-ingest --skip-packed --skip-semiglobal --pairs 65536 --pipeline-reads-total 0 --steps 2 --warmup 0"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- $BENCH > "$OUT/${TAG}_bench.log" 2>&1
cp "$OUT"/kt/bench_kernel_stats.csv "$OUT/${TAG}_bench_kernel_stats.csv" 2>/dev/null

i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pmc_$i" -o bench -- $PMCBENCH > "$OUT/pmc_$i.log" 2>&1
done

for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/big_$C" -o bench -- $BIGBENCH > "$OUT/big_$C.log" 2>&1
done

if [ -x "$R/tools/pmc_calib" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/cal_$C" -o cal -- "$R/tools/pmc_calib" > "$OUT/cal_$C.log" 2>&1
  done
fi
python "$R/tools/pmc_summary.py" "$OUT" "$TAG"
# the raw traces are bulky (every torch kernel of the run): keep the summaries only
rm -rf "$OUT"/kt "$OUT"/pmc_[0-9] "$OUT"/cal_FETCH_SIZE "$OUT"/cal_WRITE_SIZE "$OUT"/big_FETCH_SIZE "$OUT"/big_WRITE_SIZE
