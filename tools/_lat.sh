for lg in 10 14 16; do
echo "== n=2^$lg"
SONIC_DEBUG_TIMING=1 python3 bench.py --no-cpu --no-pipeline --log2n $lg --msm-log2 10 --steps 4 --warmup 2 2>&1 | grep "sonic\]" | tail -6
done
