#!/bin/bash
# Collects the measurements DESIGN.md section 5 quotes, on one MI355X box:  bash tools/collect_profiles.sh r03
# Writes gpurun_out/<tag>/...; tools/publish_profiles.sh <tag> copies the summaries into profiles/ afterwards.
# PMC passes are separate runs with one counter group each and no tracing, as MI355X_MICROARCH.md's HBM section prescribes.
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/cpu_scaling.py --log2 18 > $OUT/cpu_scaling.txt 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
./tools/microbench > $OUT/microbench.txt 2>&1
# per-category ISA budget of one bucket-walk addition at the microbenchmark's issue rates (round 6; the static count needs no GPU, the rates are this box's)
python3 tools/accum_isa.py $OUT/microbench.txt > $OUT/accum_isa.txt 2>&1
# stand-alone MSMs over small SRSs: two / four lanes per bucket against one (round 6)
{ echo "# python tools/msm_small.py: stand-alone MSMs over a SMALL SRS (one job of 2^15 / 2^16 shared buckets), scalars resident in HBM, one MSM at a time"
  echo "## two / four lanes per bucket (k_bucket_accum_split: launches of <= 131072 / <= 65536 buckets)"; python3 tools/msm_small.py 2>/dev/null | grep "d=2"
  echo "## SONIC_ACCUM_LANES=1: one lane per bucket (round 5)"; SONIC_ACCUM_LANES=1 python3 tools/msm_small.py 2>/dev/null | grep "d=2"; } > $OUT/msm_small.txt
./tools/ba_bench > $OUT/ba_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/msm_only -o t -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 5 --warmup 1 > $OUT/msm_only.json 2> $OUT/msm_only.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof -o t -- python3 bench.py --no-cpu --no-sensitivities > $OUT/bench_under_rocprof.json 2> $OUT/bench_prof.err
# the NTT product alone (roofline_ntt's kernels): per-kernel durations as the profiler sees them, and their HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ntt_only -o t -- python3 tools/ntt_time.py > $OUT/ntt_only.txt 2> $OUT/ntt_only.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/ntt_pmc_$C -o t -- python3 tools/ntt_time.py > /dev/null 2> $OUT/ntt_pmc_$C.err
done
# LDS activity and bank conflicts (SURVEY 8d's counter list): the transforms and the sort passes
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/lds_ntt -o t -- python3 tools/ntt_time.py > /dev/null 2> $OUT/lds_ntt.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/lds_msm -o t -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 3 --warmup 1 > /dev/null 2> $OUT/lds_msm.err
{
  echo "# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -- python3 tools/ntt_time.py   (per launch, summed over the chip)"
  python3 tools/pmc_lds_summary.py $(find $OUT/lds_ntt -name "*counter_collection.csv" | head -1) k_ntt_wide k_ntt_local
  echo "# ... -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 3 --warmup 1"
  python3 tools/pmc_lds_summary.py $(find $OUT/lds_msm -name "*counter_collection.csv" | head -1) k_part_scatter_staged k_part_sort k_part_hist k_border_place
} > $OUT/lds_counters.txt
# the opt-in single-pass form of the wide stages (SONIC_NTT_BIG=1; set in this shell's environment, not behind `--`): its traffic and durations
export SONIC_NTT_BIG=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ntt_only_big -o t -- python3 tools/ntt_time.py > $OUT/ntt_only_big.txt 2> $OUT/ntt_only_big.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/ntt_big_pmc_$C -o t -- python3 tools/ntt_time.py > /dev/null 2> $OUT/ntt_big_pmc_$C.err
done
unset SONIC_NTT_BIG
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o t -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 3 --warmup 1 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_SQ -o t -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 3 --warmup 1 > $OUT/pmc_SQ.json 2> $OUT/pmc_SQ.err
# VALU instruction budget of a proof (3 proofs: 1 warm-up + 2 timed, no pipeline, no stand-alone MSM leg to speak of)
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_valu -o t -- python3 bench.py --no-cpu --no-pipeline --prove-only --steps 2 --warmup 1 > $OUT/pmc_valu.json 2> $OUT/pmc_valu.err
python3 tools/valu_budget.py $(find $OUT/pmc_valu -name "*counter_collection.csv" | head -1) 3 > $OUT/valu_budget.txt 2>&1
{
  echo "# python bench.py --no-cpu --log2n <k> (10 streamed proofs after 2 warm-up, Q = 2, d = 8n): ms per proof streamed / strictly sequential"
  for lg in 10 13 14 15 16 17 18 19 20; do
    python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  d=2^$((lg+3))  streamed %.2f  sequential %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof']))"
  done
  echo "# sensitivity to Q at n = 2^18 (7 + 4Q MSMs per proof)"
  for q in 1 2 4 8; do
    python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --Q $q --msm-log2 12 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Q=$q  streamed %.2f  sequential %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof']))"
  done
} > $OUT/prove_sizes.txt
{
  echo "# python bench.py --no-cpu --msm-only --msm-log2 <k> (SRS d = 2^21: window tables c = 20, 13 windows, 2^19 shared buckets): ms per MSM and scalar-muls/s, streamed over three lanes / one at a time"
  for lg in 16 18 20 22; do
    python3 bench.py --no-cpu --no-sensitivities --msm-only --msm-log2 $lg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2^$lg  streamed %.2f ms  %.3g /s   one at a time %.2f ms  %.3g /s' % (d['msm']['ms_per_msm'], d['msm']['value'], d['msm']['sequential']['ms_per_msm'], d['msm']['sequential']['scalar_muls_per_s']))"
  done
} > $OUT/msm_sizes.txt
{
  echo "# python bench.py --msm-strong --emulate-world E --no-cpu --steps 10: one GPU doing ONE rank's share of an N = 2^22 MSM split over E ranks (UNMEASURED ON MULTI-GPU HARDWARE: device copy instead of the xGMI all-to-all)"
  for E in 2 4 8; do
    python3 bench.py --msm-strong --emulate-world $E --no-cpu --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['msm_strong']; e=d['emulated_share']; print('E=$E  whole MSM on one GPU %.3f ms   one share: kernels %.3f ms + MODELLED xGMI exchange %.3f ms = %.3f ms   speed-up %.2f (%.2f before the exchange is counted)   kernels %s' % (d['ms_per_msm'], e['ms_per_share_kernels_only'], e['exchange_model']['ms'], e['ms_per_share'], e['speedup_vs_single'], e['speedup_without_the_exchange'], {k: v for k, v in e['kernel_ms'].items() if v >= 0.02}))"
  done
} > $OUT/msm_strong_emulated.txt
# (round 5: the knobs behind round 4's bucket_tree.txt are gone; the one A/B that was repeated -- the butterfly in every group of a proof --
# is profiles/r05_prove_tree_ab.txt, measured at commit 6662339 where SONIC_PROVE_TREE still existed)
python3 tools/prove_strong.py --log2n 20 --worlds 2,3,4,6,8 --steps 3 --fit > $OUT/prove_strong_emulated.txt 2>&1
python3 tools/prove_strong.py --log2n 18 --worlds 2,4,8 --steps 3 >> $OUT/prove_strong_emulated.txt 2>&1
# one process, the C ABI, a device list: two handles standing in for two GPUs on this one (overheads of the in-process form, not scaling)
python3 bench.py --gpus 2 --in-process --devices 0,0 --no-cpu > $OUT/in_process_one_gpu.json 2> $OUT/in_process_one_gpu.err
# the protocol-shaped stand-alone MSMs alone, and the kernels of the heavy-bucket one (s(X,y): 2n of 3n + 1 scalars are two values)
python3 tools/msm_shaped.py --steps 20 > $OUT/msm_shaped.json 2> $OUT/msm_shaped.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shaped_sy -o t -- python3 tools/msm_shaped.py --which sy --no-uniform --steps 20 > /dev/null 2> $OUT/shaped_sy.err
bash tools/ab_runs.sh > $OUT/runs_ab.txt 2>&1
bash tools/ab_sym.sh > $OUT/sym_ab.txt 2>&1
./tools/mfma_bound > $OUT/mfma_bound.txt 2>&1
python3 tools/criterion_shape.py > $OUT/criterion_shape.txt 2>&1
# small proofs (round 6): one solo proof per hardware queue at n = 2^14, and how two streamed proofs hand the chip over at n = 2^16
rocprofv3 --kernel-trace --output-format csv -d $OUT/solo14 -o t -- python3 bench.py --log2n 14 --steps 3 --warmup 2 --no-cpu --no-pipeline --prove-only --strong-log2n 0 > /dev/null 2> $OUT/solo14.err
python3 tools/perqueue.py $(find $OUT/solo14 -name "*kernel_trace.csv" | head -1) 15 > $OUT/timeline_solo14.txt 2>&1
bash tools/ab_small.sh "14 16" 30 > $OUT/ab_small.txt 2>&1
python3 tools/throughput_mode.py > $OUT/throughput_mode.txt 2>&1
python3 tools/throughput_mode.py --log2n 18 --proofs 16 >> $OUT/throughput_mode.txt 2>&1
# timeline of one solo proof (no streaming): where the time of prove() goes
rocprofv3 --kernel-trace --output-format csv -d $OUT/solo -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pipeline --prove-only > $OUT/solo.json 2> $OUT/solo.err
python3 tools/timeline.py $(find $OUT/solo -name "*kernel_trace.csv" | head -1) 250 > $OUT/timeline_solo.txt 2>&1
find $OUT -name "*.csv" | head -40
