/* abi_harness.c -- the drop-in boundary exercised from plain C99, exactly as a `foreign import ccall` shim would bind it
 * (INTEGRATION.md section 3): nothing but include/sonic_hip.h, pointers and sizes.
 *
 * Runs examples/Main.hs of the reference (arithCircuitExample, examples/Main.hs:38-63: 5 linear constraints, 2 multiplication
 * gates; SRS.new with d = 25 n as bench/Main.hs:18-19; prove; verify) with z = 2, then the same through the resident prover
 * handle, the opt-in Fiat-Shamir mode, and the error contract (Protocol.hs:54-55 "Parameter d is not large enough").
 *
 *   gcc -std=c99 -pedantic -Wall -Wextra -Werror -Iinclude tests/host/abi_harness.c -Lsonic_amd/csrc -lsonic_hip -o abi_harness
 *   LD_LIBRARY_PATH=sonic_amd/csrc ./abi_harness        -> "abi_harness: OK" (exit 0); exit 77 without a GPU
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sonic_hip.h"

static void fr_small(uint8_t out[32], uint64_t v) {
  int i;
  memset(out, 0, 32);
  for (i = 0; i < 8; i++) out[i] = (uint8_t)(v >> (8 * i));
}
/* r - 1, little-endian: r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 */
static void fr_minus_one(uint8_t out[32]) {
  static const uint8_t r_be[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                   0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
  int i;
  for (i = 0; i < 32; i++) out[i] = r_be[31 - i];
  out[0] -= 1;
}

#define N 2
#define Q 5
#define D (25 * N)

static int fail(const char* what, int rc) {
  char msg[512];
  sonic_last_error(msg, sizeof msg);
  fprintf(stderr, "abi_harness: %s failed with status %d: %s\n", what, rc, msg);
  return 1;
}

int main(void) {
  uint8_t wL[Q * N * 32], wR[Q * N * 32], wO[Q * N * 32], cs[Q * 32], aL[N * 32], aR[N * 32], aO[N * 32];
  uint8_t x[32], alpha[32], tr[(8 + 2 * Q) * 32], yzs[Q * 64];
  uint8_t *proof, *proof2, *proof3, *fs_tr;
  uint8_t digest[32], seed[32];
  size_t psz = sonic_proof_size(Q);
  sonic_srs_t* srs = NULL;
  sonic_srs_t* small = NULL;
  sonic_prover_t* p = NULL;
  int rc, ok = -1, i;
  char msg[512];

  rc = sonic_init(0);
  if (rc == SONIC_ERR_NO_DEVICE) {
    sonic_last_error(msg, sizeof msg);
    fprintf(stderr, "abi_harness: SONIC_ERR_NO_DEVICE: %s\n", msg);
    return 77;
  }
  if (rc) return fail("sonic_init", rc);

  /* arithCircuitExample with z = 2: aL = (4 - z, 9 - z), aR = (9 - z, 4 - z), aO = aL * aR */
  memset(wL, 0, sizeof wL); memset(wR, 0, sizeof wR); memset(wO, 0, sizeof wO);
  fr_small(wL + 32 * (1 * N + 0), 1); fr_small(wL + 32 * (2 * N + 1), 1);
  fr_small(wR + 32 * (3 * N + 0), 1); fr_small(wR + 32 * (4 * N + 1), 1);
  fr_small(wO + 32 * (0 * N + 0), 1); fr_minus_one(wO + 32 * (0 * N + 1));
  fr_small(cs + 0, 0); fr_small(cs + 32, 2); fr_small(cs + 64, 7); fr_small(cs + 96, 7); fr_small(cs + 128, 2);
  fr_small(aL, 2); fr_small(aL + 32, 7); fr_small(aR, 7); fr_small(aR + 32, 2); fr_small(aO, 14); fr_small(aO + 32, 14);
  fr_small(x, 0x1234567u); fr_small(alpha, 0x7654321u);
  for (i = 0; i < 8 + 2 * Q; i++) fr_small(tr + 32 * i, 1000003u * (uint64_t)(i + 1) + 17);
  for (i = 0; i < Q; i++) { memcpy(yzs + 64 * i, tr + 32 * (6 + i), 32); memcpy(yzs + 64 * i + 32, tr + 32 * (6 + Q + i), 32); }

  proof = malloc(psz); proof2 = malloc(psz); proof3 = malloc(psz); fs_tr = malloc((8 + 2 * Q) * 32);
  if (!proof || !proof2 || !proof3 || !fs_tr) return 1;

  if ((rc = sonic_srs_new(D, x, alpha, &srs))) return fail("sonic_srs_new", rc);
  if (sonic_srs_d(srs) != D) return fail("sonic_srs_d", -1);
  /* prove :: SRS -> Assignment -> ArithCircuit -> m (Proof, RndOracle) */
  if ((rc = sonic_prove(srs, N, Q, wL, wR, wO, cs, aL, aR, aO, tr, proof))) return fail("sonic_prove", rc);
  /* verify :: SRS -> ArithCircuit -> Proof -> Fr -> Fr -> [(Fr, Fr)] -> Bool */
  if ((rc = sonic_verify(srs, N, Q, wL, wR, wO, cs, proof, tr + 32 * 4, tr + 32 * 5, yzs, &ok))) return fail("sonic_verify", rc);
  if (ok != 1) { fprintf(stderr, "abi_harness: verify rejected an honest proof\n"); return 1; }
  memcpy(proof2, proof, psz);
  proof2[192] ^= 1;                                   /* prA */
  rc = sonic_verify(srs, N, Q, wL, wR, wO, cs, proof2, tr + 32 * 4, tr + 32 * 5, yzs, &ok);
  if (rc || ok != 0) { fprintf(stderr, "abi_harness: verify accepted a tampered proof (status %d, accepted %d)\n", rc, ok); return 1; }

  /* the resident handle gives the same bytes, prepared or not */
  if ((rc = sonic_prover_new(srs, N, Q, wL, wR, wO, cs, &p))) return fail("sonic_prover_new", rc);
  if ((rc = sonic_prover_set_assignment(p, aL, aR, aO))) return fail("sonic_prover_set_assignment", rc);
  if ((rc = sonic_prover_prove(p, tr, proof2))) return fail("sonic_prover_prove", rc);
  if (memcmp(proof, proof2, psz)) { fprintf(stderr, "abi_harness: handle and one-shot proofs differ\n"); return 1; }
  if ((rc = sonic_prover_prepare(p))) return fail("sonic_prover_prepare", rc);
  if ((rc = sonic_prover_submit(p, tr)) || (rc = sonic_prover_collect(p, proof2))) return fail("sonic_prover_submit/collect", rc);
  if (memcmp(proof, proof2, psz)) { fprintf(stderr, "abi_harness: prepared handle gives other bytes\n"); return 1; }

  /* opt-in Fiat-Shamir: the proof carries its own challenges */
  memset(seed, 0x5a, sizeof seed);
  if ((rc = sonic_fs_circuit_digest(N, Q, wL, wR, wO, cs, digest))) return fail("sonic_fs_circuit_digest", rc);
  if ((rc = sonic_prover_prove_fs(p, digest, seed, proof3, fs_tr))) return fail("sonic_prover_prove_fs", rc);
  if ((rc = sonic_verify_fs(srs, N, Q, wL, wR, wO, cs, proof3, &ok))) return fail("sonic_verify_fs", rc);
  if (ok != 1) { fprintf(stderr, "abi_harness: verify_fs rejected an honest proof\n"); return 1; }
  if ((rc = sonic_prover_prove(p, fs_tr, proof2))) return fail("sonic_prover_prove(fs transcript)", rc);
  if (memcmp(proof3, proof2, psz)) { fprintf(stderr, "abi_harness: the reported transcript does not reproduce the Fiat-Shamir proof\n"); return 1; }
  proof3[psz - 1] ^= 1;                               /* v */
  rc = sonic_verify_fs(srs, N, Q, wL, wR, wO, cs, proof3, &ok);
  if (ok != 0) { fprintf(stderr, "abi_harness: verify_fs accepted a tampered proof (status %d)\n", rc); return 1; }
  /* ONE proof made by three ranks: each runs its share on the same handle in turn (on a node: one handle per GPU), the shares are
   * what the ranks would all-gather, sonic_proof_from_shares lays out the same bytes */
  {
    const size_t ssz = sonic_proof_share_size(Q);
    uint8_t* shares = malloc(3 * ssz);
    int r;
    if (!shares) return 1;
    for (r = 0; r < 3; r++) {
      if ((rc = sonic_prover_set_share(p, r, 3))) return fail("sonic_prover_set_share", rc);
      if ((rc = sonic_prover_prove_share(p, tr, shares + (size_t)r * ssz))) return fail("sonic_prover_prove_share", rc);
    }
    if ((rc = sonic_proof_from_shares(Q, 3, shares, tr, proof2))) return fail("sonic_proof_from_shares", rc);
    if (memcmp(proof, proof2, psz)) { fprintf(stderr, "abi_harness: the proof made of three shares differs from the proof of one GPU\n"); return 1; }
    rc = sonic_proof_from_shares(Q, 2, shares, tr, proof2);               /* a share missing: refused, not guessed */
    if (rc != SONIC_ERR_INVALID_ARG) { fprintf(stderr, "abi_harness: an incomplete set of shares gave status %d\n", rc); return 1; }
    if ((rc = sonic_prover_set_share(p, 0, 1))) return fail("sonic_prover_set_share(whole)", rc);
    free(shares);
  }
  sonic_prover_free(p);

  /* Protocol.hs:54-55: d < 7 n is refused with a status, never an abort */
  if ((rc = sonic_srs_new(7 * N - 1, x, alpha, &small))) return fail("sonic_srs_new(small)", rc);
  rc = sonic_prove(small, N, Q, wL, wR, wO, cs, aL, aR, aO, tr, proof2);
  sonic_last_error(msg, sizeof msg);
  if (rc != SONIC_ERR_D_TOO_SMALL || !strstr(msg, "not large enough")) { fprintf(stderr, "abi_harness: expected D_TOO_SMALL, got %d (%s)\n", rc, msg); return 1; }
  sonic_srs_free(small);
  sonic_srs_free(srs);
  free(proof); free(proof2); free(proof3); free(fs_tr);
  printf("abi_harness: OK (%lu proof bytes, prove + verify + handle + Fiat-Shamir + shared proof + error contract through the C ABI)\n", (unsigned long)psz);
  return 0;
}
