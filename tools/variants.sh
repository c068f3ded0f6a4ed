run() { echo "== $*"; env "$@" python bench.py --steps 4 --warmup 1 --no-cpu --log2n 20 --msm-log2 22 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streamed ms', d['ms_per_step'], 'sequential', d['sequential']['ms_per_proof'], 'msm', d['msm']['ms_per_msm'], d['int_roofline']['plan'])"; }
run SONIC_MSM_TABLE_C=20
run SONIC_MSM_TABLE_C=21
run SONIC_MSM_TABLE_C=22
