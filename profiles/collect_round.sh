#!/bin/bash
# Usage (GPU box, repo root): bash profiles/collect_round.sh <round-tag, e.g. r01>
# For every bench workload: rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W`
# -> gpurun_out/round_<tag>/<W>_kernel_stats.csv + <W>_bench.json (the JSON line of the same run);
# then the two --pmc passes (FETCH_SIZE / WRITE_SIZE, kernel trace only) of the headline.
# Copy gpurun_out/round_<tag>/* into profiles/ afterwards (gpurun_out/ is scratch).
export TMPDIR=/tmp
R=$PWD
tag=$1
out=$R/gpurun_out/round_$tag
mkdir -p $out
for w in filter_groupby_sum_i64 groupby_2key_i64 hash_agg_f64 filter_i64 filter_utf8 take_i64 take_random_i64 sort_u64 sort_f64 sort_2key compare_i64 mask_and_kleene agg_f64; do
  rm -rf $R/gpurun_out/tmp_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tmp_prof -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/${w}.log 2>&1
  cp $(ls $R/gpurun_out/tmp_prof/*/*kernel_stats.csv | head -1) $out/${tag}_${w}_kernel_stats.csv
  grep -h '^{' $out/${w}.log > $out/${tag}_${w}_bench.json
  echo "$w: $(cut -c95-200 $out/${tag}_${w}_bench.json)"
done
# variants of a workload off the uniform path: <name>:<workload>:<bench arguments>
for v in "groupby_wide_keys:filter_groupby_sum_i64:--key-bits 64 --wide-keys" "groupby_4m_groups:filter_groupby_sum_i64:--groups 4000000" \
         "sort_2key_composite:sort_2key:--value-range small" "filter_utf8_90:filter_utf8:--selectivity 0.9"; do
  name=${v%%:*}; rest=${v#*:}; w=${rest%%:*}; args=${rest#*:}
  rm -rf $R/gpurun_out/tmp_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tmp_prof -- python3 bench.py --workload $w $args --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/${name}.log 2>&1
  cp $(ls $R/gpurun_out/tmp_prof/*/*kernel_stats.csv | head -1) $out/${tag}_${name}_kernel_stats.csv
  grep -h '^{' $out/${name}.log > $out/${tag}_${name}_bench.json
  echo "$name: $(cut -c95-200 $out/${tag}_${name}_bench.json)"
done
bash profiles/run_pmc.sh $tag filter_groupby_sum_i64 > $out/${tag}_headline_hbm_pmc.txt 2>&1
cat $out/${tag}_headline_hbm_pmc.txt
bash profiles/run_pmc.sh $tag sort_u64 > $out/${tag}_sort_u64_hbm_pmc.txt 2>&1
grep PER_STEP_TOTAL $out/${tag}_sort_u64_hbm_pmc.txt
bash profiles/run_pmc.sh $tag sort_f64 > $out/${tag}_sort_f64_hbm_pmc.txt 2>&1
grep PER_STEP_TOTAL $out/${tag}_sort_f64_hbm_pmc.txt
