#!/bin/bash
# usage: bash exp/variants3.sh V1 V2 ...   (exp/libamdarrow_<V>.so); prints the scatter kernel time of each
export TMPDIR=/tmp
R=$PWD
cp arrow-1_amd/libamdarrow.so /tmp/lib_orig.so
for v in "$@"; do
  if [ "$v" != base ]; then cp exp/libamdarrow_$v.so arrow-1_amd/libamdarrow.so; else cp /tmp/lib_orig.so arrow-1_amd/libamdarrow.so; fi
  rm -rf gpurun_out/var_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$v -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline $EXTRA > gpurun_out/var_$v.log 2>&1
  python3 profiles/summarize.py "gpurun_out/var_$v/*/*kernel_stats.csv" > gpurun_out/var_$v.txt
  echo "== $v: $(grep -h '^{' gpurun_out/var_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms')" 2>/dev/null)"
  grep -E "${KPAT:-k_part_scatter|k_part_aggregate}" gpurun_out/var_$v.txt | cut -c1-60,100-140
done
cp /tmp/lib_orig.so arrow-1_amd/libamdarrow.so
