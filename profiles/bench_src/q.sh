#!/bin/bash
# quick: kernel times of the headline (or $1 workload) under rocprofv3
export TMPDIR=/tmp
R=$PWD
w=${1:-filter_groupby_sum_i64}; shift
rm -rf gpurun_out/q_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_prof -- python3 bench.py --workload $w --steps 15 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/q_prof.log 2>&1
grep -h '^{' gpurun_out/q_prof.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], round(d['ms_per_step'],4),'ms', 'frac', round(d['roofline']['frac'],4))"
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/q_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "amdarrow" in r["Name"] and int(r["Calls"]) >= 10: print("  %-58s %3s avg %8.1f min %8.1f" % (r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
