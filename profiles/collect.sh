#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash profiles/collect.sh r3 [natl5|socn5|atmos|natl1_slabs|natl1_one_slab]'):
#   natl5 (default): the default bench.py command
#       pass 1  rocprofv3 --kernel-trace --stats        -> kernel durations
#       pass 2  rocprofv3 --pmc FETCH_SIZE              (own pass, kernel-trace only)
#       pass 3  rocprofv3 --pmc WRITE_SIZE              (own pass)
#   socn5 / atmos / natl1_slabs / natl1_one_slab / natl5_one_slab_of_8 / natl5_oml: profiles/tools/run_workload.py <what> under
#       the same passes (the HBM-bound configuration BASELINE configs[2], the atmospheric channel, NAtl 1 km as eight slabs,
#       one slab of the headline basin, the mixed layer); PMC passes for socn5, natl1_slabs and natl5_oml
# then profiles/pmc_reduce.py turns the counter dumps into per-kernel HBM-side bytes per launch
# (FETCH_SIZE x2 correction for gfx950, /opt/skills/guides/MI355X_MICROARCH.md).  Results land under
# gpurun_out/<tag>_<what>/ ; copy what is to be judged into profiles/.
set -eo pipefail
TAG=${1:-prof}
WHAT=${2:-natl5}
R=$(pwd)
OUT=$R/gpurun_out/${TAG}_${WHAT}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$R"
export QGCM_BENCH_NO_SECONDARY=1   # profile the headline workload only (no mixed-layer / SOcn secondary figures)
if [ "$WHAT" = natl5 ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 > "$OUT/pmc_$C.log" 2>&1
  done
  grep '^{' "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json" || true
else
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 profiles/tools/run_workload.py $WHAT 200 > "$OUT/run.log" 2>&1
  if [ "$WHAT" = socn5 ] || [ "$WHAT" = natl1_slabs ] || [ "$WHAT" = natl5_oml ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 profiles/tools/run_workload.py $WHAT 30 > "$OUT/pmc_$C.log" 2>&1
    done
  fi
fi
cp "$OUT"/stats/*/*_kernel_stats.csv "$OUT/kernel_stats.csv"
if [ -d "$OUT/pmc_FETCH_SIZE" ]; then
  python3 profiles/pmc_reduce.py "$OUT" > "$OUT/pmc_traffic.json"
  cat "$OUT/pmc_traffic.json"
fi
head -12 "$OUT/kernel_stats.csv" | cut -c1-150
