#!/bin/bash
export TMPDIR=/tmp
run() {  # label, log2n, env...
  local label=$1 lg=$2; shift 2
  env "$@" python3 bench.py --log2n $lg --steps 30 --warmup 4 --no-cpu --prove-only --strong-log2n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  %-34s streamed %6.2f  sequential %6.2f' % ('$label', d['ms_per_step'], d['sequential']['ms_per_proof']))"
}
for rep in 1 2; do
for lg in 16 14 10; do
  for q in 6 4 3 2 1; do run "SONIC_FUSED_LANES=$q" $lg SONIC_FUSED_LANES=$q; done
done
done
python3 tools/throughput_mode.py --depths 2 --distinct 0 2>&1 | tail -1
python3 - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sonic_amd
from sonic_amd import _lib
from util import big_circuit, rand_fr_array
L = _lib.lib(); _lib.check(L.sonic_init(0))
n, Q = 1 << 16, 2
rng = np.random.default_rng(0)
x = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
al = int.from_bytes(rand_fr_array(rng, 1)[0].tobytes(), "little") | 1
srs = sonic_amd.SRS.new(8 * n, x, al)
c = big_circuit(1, n, Q, None)
circuit = sonic_amd.ArithCircuit(sonic_amd.GateWeights(c["wL"], c["wR"], c["wO"]), c["cs"])
asg = sonic_amd.Assignment(c["aL"], c["aR"], c["aO"])
trs = [rand_fr_array(rng, 8 + 2 * Q) for _ in range(68)]
for t in trs: t[:, 0] |= 1
for nh in (2, 3):
    hs = [sonic_amd.Prover(srs, circuit, prepare=True) for _ in range(nh)]
    for h in hs: h.set_assignment(asg)
    sonic_amd.prove_batch(hs, trs[:4])
    for mode in ("resident", "per-proof"):
        L.sonic_device_sync(); t0 = time.perf_counter()
        out = sonic_amd.prove_batch(hs, trs[4:], assignments=None if mode == "resident" else [asg] * 64)
        dt = time.perf_counter() - t0
        print(f"sonic_prove_batch {nh} handles, assignment {mode}: 64 proofs in {dt*1e3:.1f} ms -> {64/dt:.1f} proofs/s")
    for h in hs: h.close()
PY
