#!/bin/bash
# Collect the profiles/ artefacts of one round on the GPU box (run through gpurun):
#   tools/collect_profiles.sh <tag>        e.g. r01
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command  -> <tag>_bench_kernel_stats.csv
# 2. separate PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains)
#                                                                     -> <tag>_pmc_traffic.json
# 3. a calibration of the two counters against kernels with known traffic -> <tag>_pmc_calibration.txt
# Everything lands in gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --skip-cpu --skip-pipeline"  # one launch shape per kernel: the seed-and-extend leg reuses K1/K2/K5

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- $BENCH > "$OUT/${TAG}_bench.log" 2>&1
cp "$OUT"/kt/bench_kernel_stats.csv "$OUT/${TAG}_bench_kernel_stats.csv" 2>/dev/null

for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o bench -- $BENCH --steps 2 --warmup 0 > "$OUT/pmc_$C.log" 2>&1
done

if [ -x "$R/tools/pmc_calib" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/cal_$C" -o cal -- "$R/tools/pmc_calib" > "$OUT/cal_$C.log" 2>&1
  done
fi
python "$R/tools/pmc_summary.py" "$OUT" "$TAG"
# the raw traces are bulky (every torch kernel of the run): keep the summaries only
rm -rf "$OUT"/kt "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/cal_FETCH_SIZE "$OUT"/cal_WRITE_SIZE
