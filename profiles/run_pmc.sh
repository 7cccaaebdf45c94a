#!/bin/bash
# Usage (GPU box, repo root): bash profiles/run_pmc.sh <tag> <workload> [bench args]
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with kernel trace only, as
# MI355X_MICROARCH.md prescribes; prints KB per dispatch per kernel (FETCH_SIZE doubled: gfx950
# reports half of the streamed bytes).
export TMPDIR=/tmp
R=$PWD
tag=$1; w=$2; shift 2
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${tag}_${w}_$c  # (a directory left by an earlier collection of the same tag would be summed in)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_${w}_$c -- python3 bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > gpurun_out/pmc_${tag}_${w}_$c.log 2>&1
done
python3 profiles/pmc_summary.py gpurun_out/pmc_${tag}_${w}_FETCH_SIZE gpurun_out/pmc_${tag}_${w}_WRITE_SIZE
