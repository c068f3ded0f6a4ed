export TMPDIR=/tmp
O=gpurun_out/r06_diag4; mkdir -p $O
for lg in 14 16; do
  rocprofv3 --kernel-trace --output-format csv -d $O/str$lg -o t -- python3 bench.py --log2n $lg --steps 8 --warmup 2 --no-cpu --prove-only --strong-log2n 0 > $O/str$lg.json 2> $O/str$lg.err
  cp $(find $O/str$lg -name "*kernel_trace.csv" | head -1) $O/str${lg}_trace.csv; rm -rf $O/str$lg
  SONIC_DEBUG_TIMING=1 python3 bench.py --log2n $lg --steps 8 --warmup 2 --no-cpu --prove-only --strong-log2n 0 > $O/t$lg.json 2> $O/t$lg.err
done
