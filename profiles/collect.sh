#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash profiles/collect.sh r2x'):
#   pass 1  rocprofv3 --kernel-trace --stats        of the default bench.py command -> kernel durations
#   pass 2  rocprofv3 --pmc FETCH_SIZE              (own pass, kernel-trace only)
#   pass 3  rocprofv3 --pmc WRITE_SIZE              (own pass)
# then profiles/pmc_reduce.py turns the counter dumps into per-kernel HBM-side bytes per launch
# (FETCH_SIZE x2 correction for gfx950, /opt/skills/guides/MI355X_MICROARCH.md).  Results land under
# gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -eo pipefail
TAG=${1:-prof}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$R"
export QGCM_BENCH_NO_SECONDARY=1   # profile the headline workload only (no mixed-layer / SOcn secondary figures)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 > "$OUT/pmc_$C.log" 2>&1
done
python3 profiles/pmc_reduce.py "$OUT" > "$OUT/pmc_traffic.json"
cp "$OUT"/stats/*/*_kernel_stats.csv "$OUT/kernel_stats.csv"
grep '^{' "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json" || true
cat "$OUT/pmc_traffic.json"
