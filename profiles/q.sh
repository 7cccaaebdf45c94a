#!/bin/bash
# quick profile of one workload: bash profiles/q.sh <tag> <workload> [bench args...]  -> gpurun_out/q_<tag>_<workload>.{log,csv}
export TMPDIR=/tmp
R=$PWD
tag=$1; w=$2; shift 2
rm -rf $R/gpurun_out/tmp_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tmp_prof -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/q_${tag}_${w}.log 2>&1
cp $(ls $R/gpurun_out/tmp_prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/q_${tag}_${w}.csv
grep -h '^{' $R/gpurun_out/q_${tag}_${w}.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'ms_per_step', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'frac', round(d['roofline']['frac'],4))"
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$R/gpurun_out/q_${tag}_${w}.csv')) if 'amdarrow' in r['Name']]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:12]:
    print('%9.1f us x %4s  %s' % (float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:110]))
PY
