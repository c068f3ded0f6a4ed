#!/bin/bash
# Copies the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/ under the round's prefix, and derives
# the two JSON files bench.py reads (kernel model, HBM traffic per kernel).   bash tools/publish_profiles.sh r03
TAG=${1:-r06}
IN=gpurun_out/$TAG
P=profiles
for f in bench.json lds_counters.txt microbench.txt ba_bench.txt in_process_one_gpu.json prove_strong_emulated.txt mfma_bound.txt criterion_shape.txt msm_only.json bench_under_rocprof.json prove_sizes.txt msm_sizes.txt msm_strong_emulated.txt throughput_mode.txt timeline_solo.txt valu_budget.txt cpu_scaling.txt msm_shaped.json runs_ab.txt sym_ab.txt accum_isa.txt msm_small.txt timeline_solo14.txt; do
  [ -f $IN/$f ] && cp $IN/$f $P/${TAG}_$f
done
cp $(find $IN/msm_only -name "*kernel_stats.csv" | head -1) $P/${TAG}_msm_only_kernel_stats.csv
cp $(find $IN/bench_prof -name "*kernel_stats.csv" | head -1) $P/${TAG}_bench_kernel_stats.csv
[ -d $IN/shaped_sy ] && cp $(find $IN/shaped_sy -name "*kernel_stats.csv" | head -1) $P/${TAG}_msm_shaped_sy_kernel_stats.csv
[ -d $IN/ntt_only ] && cp $(find $IN/ntt_only -name "*kernel_stats.csv" | head -1) $P/${TAG}_ntt_only_kernel_stats.csv && cp $IN/ntt_only.txt $P/${TAG}_ntt_only.txt
for C in FETCH_SIZE WRITE_SIZE; do
  [ -d $IN/ntt_pmc_$C ] && cp $(find $IN/ntt_pmc_$C -name "*counter_collection.csv" | head -1) $P/${TAG}_ntt_pmc_${C}_counter_collection.csv
done
if [ -d $IN/ntt_only_big ]; then
  cp $(find $IN/ntt_only_big -name "*kernel_stats.csv" | head -1) $P/${TAG}_ntt_only_big_kernel_stats.csv && cp $IN/ntt_only_big.txt $P/${TAG}_ntt_only_big.txt
  for C in FETCH_SIZE WRITE_SIZE; do cp $(find $IN/ntt_big_pmc_$C -name "*counter_collection.csv" | head -1) $P/${TAG}_ntt_big_pmc_${C}_counter_collection.csv; done
  python3 tools/pmc_ntt_summary.py $P/${TAG}_ntt_big_pmc_FETCH_SIZE_counter_collection.csv $P/${TAG}_ntt_big_pmc_WRITE_SIZE_counter_collection.csv $P/${TAG}_ntt_only_big_kernel_stats.csv > $P/${TAG}_pmc_ntt_big.json
fi
[ -d $IN/ntt_pmc_WRITE_SIZE ] && python3 tools/pmc_ntt_summary.py $P/${TAG}_ntt_pmc_FETCH_SIZE_counter_collection.csv $P/${TAG}_ntt_pmc_WRITE_SIZE_counter_collection.csv $P/${TAG}_ntt_only_kernel_stats.csv > $P/${TAG}_pmc_ntt.json
for C in FETCH_SIZE WRITE_SIZE SQ; do
  cp $(find $IN/pmc_$C -name "*counter_collection.csv" | head -1) $P/${TAG}_pmc_${C}_counter_collection.csv
done
read c w s < <(python3 -c "import json; j=json.load(open('$IN/bench.json'))['int_roofline']['plan'] if json.load(open('$IN/bench.json')).get('int_roofline',{}).get('plan') else {'window_bits':20,'windows':13,'bucket_sets':1}; print(j['window_bits'], j['windows'], j['bucket_sets'])")
python3 tools/pmc_summary.py $P/${TAG}_pmc_FETCH_SIZE_counter_collection.csv $P/${TAG}_pmc_WRITE_SIZE_counter_collection.csv 1048576 $c $w $s > $P/${TAG}_pmc_msm.json
python3 tools/kernel_model.py $TAG $P/${TAG}_microbench.txt > /dev/null
[ -f $IN/ab_small.txt ] && cp $IN/ab_small.txt $P/${TAG}_ab_small_final.txt
ls -la $P | grep ${TAG}_
