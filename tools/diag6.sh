export TMPDIR=/tmp
O=gpurun_out/r06_diag6; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() {  # label, log2n, env...
  local label=$1 lg=$2; shift 2
  env "$@" python3 bench.py --log2n $lg --steps 30 --warmup 4 --no-cpu --prove-only --strong-log2n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  %-34s streamed %6.2f  sequential %6.2f' % ('$label', d['ms_per_step'], d['sequential']['ms_per_proof']))"
}
for rep in 1 2; do
for lg in 10 13 14 15 16; do run default $lg A=1; done
run "c=17" 15 SONIC_MSM_TABLE_C=17
run "c=16" 15 SONIC_MSM_TABLE_C=16
run "c=15" 13 SONIC_MSM_TABLE_C=15
done
python3 tools/criterion_shape.py 2>&1 | tail -2
python3 tools/throughput_mode.py --depths 2 2>&1 | tail -2
for E in 8; do
  for v in 1 0; do
  SONIC_SUM_SLICES4=$v python3 bench.py --msm-strong --emulate-world $E --no-cpu --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['msm_strong']; e=d['emulated_share']; print('E=$E sum4=$v whole MSM on one GPU %.3f ms   one share: kernels %.3f ms + MODELLED xGMI exchange %.3f ms = %.3f ms   speed-up %.2f (%.2f before the exchange is counted)   kernels %s' % (d['ms_per_msm'], e['ms_per_share_kernels_only'], e['exchange_model']['ms'], e['ms_per_share'], e['speedup_vs_single'], e['speedup_without_the_exchange'], {k: v for k, v in e['kernel_ms'].items() if v >= 0.02}))"
  done
done
rocprofv3 --kernel-trace --output-format csv -d $O/str16 -o t -- python3 bench.py --log2n 16 --steps 8 --warmup 2 --no-cpu --prove-only --strong-log2n 0 > $O/str16.json 2> $O/str16.err
cp $(find $O/str16 -name "*kernel_trace.csv" | head -1) $O/str16_trace.csv; rm -rf $O/str16
