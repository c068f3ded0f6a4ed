export TMPDIR=/tmp
O=gpurun_out/r06_diag3; mkdir -p $O
for lg in 14 16; do
  rocprofv3 --kernel-trace --output-format csv -d $O/solo$lg -o t -- python3 bench.py --log2n $lg --steps 3 --warmup 2 --no-cpu --no-pipeline --prove-only --strong-log2n 0 > $O/solo$lg.json 2> $O/solo$lg.err
  python3 tools/timeline.py $(find $O/solo$lg -name "*kernel_trace.csv" | head -1) 40 > $O/timeline_solo$lg.txt 2>&1
  SONIC_DEBUG_TIMING=1 python3 bench.py --log2n $lg --steps 5 --warmup 2 --no-cpu --no-pipeline --prove-only --strong-log2n 0 > $O/timing$lg.json 2> $O/timing$lg.err
done
rocprofv3 --kernel-trace --output-format csv -d $O/crit -o t -- python3 tools/criterion_shape.py > $O/crit.txt 2> $O/crit.err
cp $(find $O/crit -name "*kernel_trace.csv" | head -1) $O/crit_trace.csv
for lg in 14 16; do cp $(find $O/solo$lg -name "*kernel_trace.csv" | head -1) $O/solo${lg}_trace.csv; done
rm -rf $O/solo14 $O/solo16 $O/crit
ls -la $O
