#!/bin/bash
export TMPDIR=/tmp
run() {  # label, log2n, env...
  local label=$1 lg=$2; shift 2
  env "$@" python3 bench.py --log2n $lg --steps 30 --warmup 4 --no-cpu --prove-only --strong-log2n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  %-34s streamed %6.2f  sequential %6.2f' % ('$label', d['ms_per_step'], d['sequential']['ms_per_proof']))"
}
for rep in 1 2; do
for lg in 16 14; do
  for r in 0 8 16 24 32 48 64; do run "accum CU reserve $r" $lg SONIC_ACCUM_CU_RESERVE=$r; done
done
done
for v in 1 0; do
SONIC_SUM_SLICES2=$v python3 bench.py --msm-strong --emulate-world 8 --no-cpu --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['msm_strong']; e=d['emulated_share']; print('E=8 sum2=$v whole MSM %.3f ms   share kernels %.3f + exchange %.3f = %.3f ms   speed-up %.2f   term-range mode %.3f ms (%.2f)  kernels %s' % (d['ms_per_msm'], e['ms_per_share_kernels_only'], e['exchange_model']['ms'], e['ms_per_share'], e['speedup_vs_single'], e['term_range_mode']['ms_per_share'], e['term_range_mode']['speedup_vs_single'], {k: v for k, v in e['kernel_ms'].items() if v >= 0.02}))"
done
