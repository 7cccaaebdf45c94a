#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> <workload> [more workloads]
# Writes rocprofv3 --kernel-trace --stats output under gpurun_out/prof_<tag>_<workload>/ and the bench JSON
# line to gpurun_out/bench_<tag>_<workload>.log.
export TMPDIR=/tmp
R=$PWD
tag=$1; shift
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${tag}_$w.log 2>&1
  grep -h '^{' gpurun_out/bench_${tag}_$w.log | cut -c1-200
done
