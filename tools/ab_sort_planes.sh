#!/bin/bash
# A/B of the split-plane record layout in k_part_scatter_staged: kernel times of the stand-alone N = 2^20 MSM and the
# LDS counters of the sort's kernels, both layouts on one box.   bash tools/ab_sort_planes.sh > profiles/r06_sort_planes_ab.txt
export TMPDIR=/tmp
O=gpurun_out/r06_planes; mkdir -p $O
for v in 0 1; do
  export SONIC_SORT_PLANES=$v
  echo "## SONIC_SORT_PLANES=$v ($([ $v = 1 ] && echo 'two planes of 4-byte slots' || echo '8-byte records: the default'))"
  python3 bench.py --no-cpu --no-sensitivities --msm-only --msm-log2 20 --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['msm']; k=m['sequential']['kernel_ms']; print('N=2^20  streamed %.3f ms  one at a time %.3f ms   sort kernels (ms per MSM): %s' % (m['ms_per_msm'], m['sequential']['ms_per_msm'], {a: b for a, b in k.items() if 'part' in a or 'border' in a or 'scan' in a}))"
  rm -rf $O/lds$v
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/lds$v -o t -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/lds$v.err
  python3 tools/pmc_lds_summary.py $(find $O/lds$v -name "*counter_collection.csv" | head -1) k_part_scatter_staged k_part_sort k_part_hist k_border_place
done
unset SONIC_SORT_PLANES
