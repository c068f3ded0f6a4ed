/* multi_harness.c -- N GPUs from ONE host process through nothing but include/sonic_hip.h (plain C99, as a `foreign import ccall`
 * shim binds it: INTEGRATION.md section 6).  The reference's prove is one pure call in one process (src/Sonic/Protocol.hs:47-52);
 * sonic_prove_shared / sonic_prove_batch / sonic_msm_g1_srs_multi keep it that way on a node of GPUs.
 *
 *   multi_harness <case file> <device list, e.g. 0,0,0 or 0,1,2,3>
 *
 * The same ordinal may appear several times: the handles then share a GPU (how the one-GPU boxes of the pool drive this), every code
 * path -- one host thread per handle, shares combined on the host, peer copies of bucket ranges -- is the one a node runs.
 *
 * case file (written by tests/test_gpu_multi.py):
 *   i64 n, Q, d, K | x, alpha (32 B each) | wL, wR, wO (Q n x 32 B each) | cs (Q x 32) | aL, aR, aO (n x 32 each) |
 *   K transcripts ((8 + 2Q) x 32 each) | the proof of transcript 0 as the CPU oracle makes it (sonic_proof_size(Q) bytes)
 *
 * Checks, byte for byte:
 *   1. sonic_prove on device[0] alone                                        == the oracle's proof
 *   2. sonic_prove_shared over all devices (replicas made by sonic_srs_new_on and by sonic_srs_replicate), unprepared and prepared,
 *      then over the first two handles only (the plan changes), then over one                == the oracle's proof
 *   3. sonic_prove_batch: K proofs over all handles, resident assignment and per-proof assignments == sonic_prover_prove one by one;
 *      sonic_prove_many: K whole statements (circuit + assignment + transcript each) over all replicas    == the same proofs
 *   4. sonic_msm_g1_srs_multi by term range and by bucket range (host scalars), sonic_msm_g1_srs_multi_dev (resident slices)
 *                                                                                             == sonic_msm_g1_srs on device[0]
 *   5. the error contract: unknown ordinal, handles of different circuits, a handle named twice, a lane and an SRS on different GPUs
 *      (only with two distinct ordinals), the ABI version
 * Prints "multi_harness: OK ..." and exits 0; exit 77 without a GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sonic_hip.h"

#define MAXDEV 16

static int fail(const char* what, int rc) {
  char msg[512];
  sonic_last_error(msg, sizeof msg);
  fprintf(stderr, "multi_harness: %s failed with status %d: %s\n", what, rc, msg);
  return 1;
}
static uint8_t* rd(FILE* f, size_t bytes) {
  uint8_t* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "multi_harness: case file too short\n"); exit(2); }
  return p;
}

int main(int argc, char** argv) {
  int64_t hdr[4], n, Q, d, K;
  uint8_t *x, *alpha, *wL, *wR, *wO, *cs, *aL, *aR, *aO, *trs, *want, *got, *batch, *batch2, *rep_aL, *rep_aR, *rep_aO, *scal;
  int dev[MAXDEV], world = 0, rc, i, ndev = 0, distinct = 0;
  sonic_srs_t* srs[MAXDEV];
  sonic_prover_t* prv[MAXDEV];
  sonic_prover_t* prv2[MAXDEV];
  size_t psz, tsz, asz;
  FILE* f;
  char* tok;
  char msg[512];

  if (argc < 3) { fprintf(stderr, "usage: multi_harness <case file> <device list>\n"); return 2; }
  for (tok = strtok(argv[2], ","); tok && world < MAXDEV; tok = strtok(NULL, ",")) dev[world++] = atoi(tok);
  if (world < 1) return 2;
  rc = sonic_init(dev[0]);
  if (rc == SONIC_ERR_NO_DEVICE) { sonic_last_error(msg, sizeof msg); fprintf(stderr, "multi_harness: SONIC_ERR_NO_DEVICE: %s\n", msg); return 77; }
  if (rc) return fail("sonic_init", rc);
  if (sonic_abi_version() != SONIC_ABI_VERSION) { fprintf(stderr, "multi_harness: library ABI %d, header %d\n", sonic_abi_version(), SONIC_ABI_VERSION); return 1; }
  if ((rc = sonic_device_count(&ndev)) || ndev < 1) return fail("sonic_device_count", rc);
  for (i = 0; i < world; i++) {
    if (dev[i] < 0 || dev[i] >= ndev) { fprintf(stderr, "multi_harness: device %d not present (%d devices)\n", dev[i], ndev); return 2; }
    if (dev[i] != dev[0]) distinct = 1;
  }

  f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "multi_harness: cannot open %s\n", argv[1]); return 2; }
  if (fread(hdr, 8, 4, f) != 4) return 2;
  n = hdr[0]; Q = hdr[1]; d = hdr[2]; K = hdr[3];
  psz = sonic_proof_size(Q); tsz = (size_t)(8 + 2 * Q) * 32; asz = (size_t)n * 32;
  x = rd(f, 32); alpha = rd(f, 32);
  wL = rd(f, (size_t)(Q * n) * 32); wR = rd(f, (size_t)(Q * n) * 32); wO = rd(f, (size_t)(Q * n) * 32); cs = rd(f, (size_t)Q * 32);
  aL = rd(f, asz); aR = rd(f, asz); aO = rd(f, asz);
  trs = rd(f, tsz * (size_t)K);
  want = rd(f, psz);
  fclose(f);
  got = malloc(psz); batch = malloc(psz * (size_t)K); batch2 = malloc(psz * (size_t)K);
  if (!got || !batch || !batch2 || K < 1) return 2;

  /* replicas: even positions by SRS.new on their device, odd positions copied from replica 0 device to device */
  for (i = 0; i < world; i++) {
    if (i == 0 || (i & 1) == 0) { if ((rc = sonic_srs_new_on(dev[i], d, x, alpha, &srs[i]))) return fail("sonic_srs_new_on", rc); }
    else if ((rc = sonic_srs_replicate(srs[0], dev[i], &srs[i]))) return fail("sonic_srs_replicate", rc);
    if (sonic_srs_device(srs[i]) != dev[i] || sonic_srs_d(srs[i]) != d) { fprintf(stderr, "multi_harness: replica %d reports device %d, d %ld\n", i, sonic_srs_device(srs[i]), (long)sonic_srs_d(srs[i])); return 1; }
  }

  /* 1. one GPU alone */
  if ((rc = sonic_prove(srs[0], n, Q, wL, wR, wO, cs, aL, aR, aO, trs, got))) return fail("sonic_prove", rc);
  if (memcmp(got, want, psz)) { fprintf(stderr, "multi_harness: the one-GPU proof differs from the oracle's\n"); return 1; }

  /* 2. ONE proof over all handles */
  for (i = 0; i < world; i++) {
    if ((rc = sonic_prover_new(srs[i], n, Q, wL, wR, wO, cs, &prv[i]))) return fail("sonic_prover_new", rc);
    if ((rc = sonic_prover_set_assignment(prv[i], aL, aR, aO))) return fail("sonic_prover_set_assignment", rc);
    if (sonic_prover_device(prv[i]) != dev[i]) { fprintf(stderr, "multi_harness: prover %d on device %d\n", i, sonic_prover_device(prv[i])); return 1; }
  }
  memset(got, 0, psz);
  if ((rc = sonic_prove_shared(prv, world, trs, got))) return fail("sonic_prove_shared", rc);
  if (memcmp(got, want, psz)) { fprintf(stderr, "multi_harness: the proof shared by %d handles differs from the oracle's\n", world); return 1; }
  for (i = 0; i < world; i++) if ((rc = sonic_prover_prepare(prv[i]))) return fail("sonic_prover_prepare", rc);
  memset(got, 0, psz);
  if ((rc = sonic_prove_shared(prv, world, trs, got))) return fail("sonic_prove_shared(prepared)", rc);
  if (memcmp(got, want, psz)) { fprintf(stderr, "multi_harness: the shared proof of prepared handles differs\n"); return 1; }
  if (world > 2) {
    memset(got, 0, psz);
    if ((rc = sonic_prove_shared(prv, 2, trs, got))) return fail("sonic_prove_shared(2)", rc);
    if (memcmp(got, want, psz)) { fprintf(stderr, "multi_harness: the proof shared by the first two handles differs\n"); return 1; }
  }
  memset(got, 0, psz);
  if ((rc = sonic_prove_shared(prv, 1, trs, got))) return fail("sonic_prove_shared(1)", rc);       /* back to a whole proof on one handle */
  if (memcmp(got, want, psz)) { fprintf(stderr, "multi_harness: sonic_prove_shared over one handle differs\n"); return 1; }

  /* 3. K proofs over all handles (a handle that ran a share is taken out of share mode first) */
  for (i = 0; i < world; i++) if ((rc = sonic_prover_set_share(prv[i], 0, 1))) return fail("sonic_prover_set_share(whole)", rc);
  for (i = 0; i < (int)K; i++)
    if ((rc = sonic_prover_prove(prv[0], trs + tsz * (size_t)i, batch2 + psz * (size_t)i))) return fail("sonic_prover_prove", rc);
  if (memcmp(batch2, want, psz)) { fprintf(stderr, "multi_harness: proof 0 of the sequential list differs from the oracle's\n"); return 1; }
  memset(batch, 0, psz * (size_t)K);
  if ((rc = sonic_prove_batch(prv, world, K, NULL, NULL, NULL, trs, batch, NULL))) return fail("sonic_prove_batch", rc);
  if (memcmp(batch, batch2, psz * (size_t)K)) { fprintf(stderr, "multi_harness: the batch over %d handles differs from the proofs made one by one\n", world); return 1; }
  rep_aL = malloc(asz * (size_t)K); rep_aR = malloc(asz * (size_t)K); rep_aO = malloc(asz * (size_t)K);
  if (!rep_aL || !rep_aR || !rep_aO) return 2;
  for (i = 0; i < (int)K; i++) { memcpy(rep_aL + asz * (size_t)i, aL, asz); memcpy(rep_aR + asz * (size_t)i, aR, asz); memcpy(rep_aO + asz * (size_t)i, aO, asz); }
  {
    int* st = malloc(sizeof(int) * (size_t)K);
    if (!st) return 2;
    memset(batch, 0, psz * (size_t)K);
    if ((rc = sonic_prove_batch(prv, world, K, rep_aL, rep_aR, rep_aO, trs, batch, st))) return fail("sonic_prove_batch(assignments)", rc);
    for (i = 0; i < (int)K; i++) if (st[i]) { fprintf(stderr, "multi_harness: batch status[%d] = %d\n", i, st[i]); return 1; }
    if (memcmp(batch, batch2, psz * (size_t)K)) { fprintf(stderr, "multi_harness: the batch with per-proof assignments differs\n"); return 1; }
    /* a bad statement in the list is reported for that proof, the others are still made */
    if (K >= 2) {
      memset(rep_aL + asz, 0xff, 32);                                   /* proof 1: aL[0] >= r */
      memset(batch, 0, psz * (size_t)K);
      rc = sonic_prove_batch(prv, world, K, rep_aL, rep_aR, rep_aO, trs, batch, st);
      if (rc != SONIC_ERR_BAD_ENCODING || st[1] != SONIC_ERR_BAD_ENCODING || st[0] != SONIC_OK || memcmp(batch, batch2, psz)) {
        fprintf(stderr, "multi_harness: a non-canonical assignment in proof 1 gave status %d (per proof: %d, %d)\n", rc, st[0], st[1]); return 1; }
      if ((rc = sonic_prover_set_assignment(prv[1 % world], aL, aR, aO))) return fail("sonic_prover_set_assignment(restore)", rc);
    }
    free(st);
  }

  /* 3b. K statements handed over whole -- circuit, assignment, transcript per proof, as the reference's prove takes them -- spread over the
   * replicas by sonic_prove_many (here the K statements share their circuit and differ in their transcripts) */
  {
    sonic_statement_t* sts = malloc(sizeof(sonic_statement_t) * (size_t)K);
    int* st = malloc(sizeof(int) * (size_t)K);
    if (!sts || !st) return 2;
    for (i = 0; i < (int)K; i++) {
      sts[i].wL = wL; sts[i].wR = wR; sts[i].wO = wO; sts[i].cs = cs; sts[i].aL = aL; sts[i].aR = aR; sts[i].aO = aO;
      sts[i].transcript = trs + tsz * (size_t)i;
    }
    memset(batch, 0, psz * (size_t)K);
    if ((rc = sonic_prove_many((const sonic_srs_t* const*)srs, world, n, Q, sts, K, batch, st))) return fail("sonic_prove_many", rc);
    for (i = 0; i < (int)K; i++) if (st[i]) { fprintf(stderr, "multi_harness: sonic_prove_many status[%d] = %d\n", i, st[i]); return 1; }
    if (memcmp(batch, batch2, psz * (size_t)K)) { fprintf(stderr, "multi_harness: sonic_prove_many over %d replicas differs from the proofs made one by one\n", world); return 1; }
    free(sts); free(st);
  }

  /* 4. ONE MSM over all replicas */
  {
    const int64_t N = 2 * d < 50000 ? 2 * d : 50000;
    uint8_t one[96], m0[96], m1[96], m2[96];
    uint64_t lcg = 0x9e3779b97f4a7c15ull;
    int64_t e0s[MAXDEV], ns[MAXDEV];
    void* dsl[MAXDEV];
    const void* cdsl[MAXDEV];
    int64_t j;
    scal = malloc(32 * (size_t)N);
    if (!scal) return 2;
    for (j = 0; j < 32 * N; j++) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; scal[j] = (uint8_t)(lcg >> 56); }
    for (j = 0; j < N; j++) scal[32 * j + 31] &= 0x3f;                 /* < 2^254 < r: canonical */
    memset(scal + 32 * 5, 0, 32);                                       /* a zero scalar ... */
    memcpy(scal + 32 * 7, scal + 32 * 6, 32);                           /* ... and a repeated one */
    if ((rc = sonic_msm_g1_srs(srs[0], 1, -d + 1, scal, N - 1 > d - 1 ? d - 1 : N - 1, one))) return fail("sonic_msm_g1_srs", rc);
    if ((rc = sonic_msm_g1_srs_multi((const sonic_srs_t* const*)srs, world, 1, -d + 1, scal, N - 1 > d - 1 ? d - 1 : N - 1, 0, m0))) return fail("sonic_msm_g1_srs_multi(term ranges)", rc);
    if (memcmp(one, m0, 96)) { fprintf(stderr, "multi_harness: the term-range MSM over %d replicas differs from one GPU's\n", world); return 1; }
    if ((rc = sonic_msm_g1_srs_multi((const sonic_srs_t* const*)srs, world, 1, -d + 1, scal, N - 1 > d - 1 ? d - 1 : N - 1, 1, m1))) return fail("sonic_msm_g1_srs_multi(bucket ranges)", rc);
    if (memcmp(one, m1, 96)) { fprintf(stderr, "multi_harness: the bucket-range MSM over %d replicas differs from one GPU's\n", world); return 1; }
    /* resident slices, uneven on purpose (the last replica gets the remainder, the first an empty slice when there are three or more) */
    {
      const int64_t total = N - 1 > d - 1 ? d - 1 : N - 1;
      int64_t at = 0;
      for (i = 0; i < world; i++) {
        int64_t cnt = (world >= 3 && i == 0) ? 0 : (i == world - 1 ? total - at : total / world);
        e0s[i] = -d + 1 + at; ns[i] = cnt; dsl[i] = NULL;
        if ((rc = sonic_dev_alloc_on(dev[i], (size_t)(cnt > 0 ? cnt : 1) * 32, &dsl[i]))) return fail("sonic_dev_alloc_on", rc);
        if (cnt > 0 && (rc = sonic_dev_upload(dsl[i], scal + 32 * at, (size_t)cnt * 32))) return fail("sonic_dev_upload", rc);
        cdsl[i] = dsl[i];
        at += cnt;
      }
      for (i = 0; i < 2; i++) {
        if ((rc = sonic_msm_g1_srs_multi_dev((const sonic_srs_t* const*)srs, world, 1, e0s, cdsl, ns, i, m2))) return fail("sonic_msm_g1_srs_multi_dev", rc);
        if (memcmp(one, m2, 96)) { fprintf(stderr, "multi_harness: the MSM over resident slices (mode %d) differs from one GPU's\n", i); return 1; }
      }
      for (i = 0; i < world; i++) sonic_dev_free(dsl[i]);
    }
    /* a non-canonical scalar in one rank's slice is that call's status */
    memset(scal + 32 * (size_t)((N - 1 > d - 1 ? d - 1 : N - 1) - 1), 0xff, 32);
    for (i = 0; i < 2; i++) {
      rc = sonic_msm_g1_srs_multi((const sonic_srs_t* const*)srs, world, 1, -d + 1, scal, N - 1 > d - 1 ? d - 1 : N - 1, i, m0);
      if (rc != SONIC_ERR_BAD_ENCODING) { fprintf(stderr, "multi_harness: a non-canonical scalar gave status %d in mode %d\n", rc, i); return 1; }
    }
    free(scal);
  }

  /* 5. the error contract */
  {
    sonic_srs_t* bad = NULL;
    rc = sonic_srs_new_on(ndev, d, x, alpha, &bad);
    if (rc != SONIC_ERR_INVALID_ARG) { fprintf(stderr, "multi_harness: SRS.new on device %d of %d gave status %d\n", ndev, ndev, rc); return 1; }
    if (world >= 2) {
      prv2[0] = prv[0]; prv2[1] = prv[0];
      rc = sonic_prove_shared(prv2, 2, trs, got);
      if (rc != SONIC_ERR_INVALID_ARG) { fprintf(stderr, "multi_harness: a handle named twice gave status %d\n", rc); return 1; }
      if (n > 1) {
        sonic_prover_t* other = NULL;
        if ((rc = sonic_prover_new(srs[1], n - 1, Q, wL, wR, wO, cs, &other))) return fail("sonic_prover_new(n - 1)", rc);
        prv2[0] = prv[0]; prv2[1] = other;
        rc = sonic_prove_shared(prv2, 2, trs, got);
        if (rc != SONIC_ERR_INVALID_ARG) { fprintf(stderr, "multi_harness: handles of different circuits gave status %d\n", rc); return 1; }
        sonic_prover_free(other);
      }
    }
    if (distinct) {
      /* a lane serves the SRS handles of its own GPU */
      sonic_msm_lane_t* lane = NULL;
      void* dsc = NULL;
      int other = 0;
      for (i = 1; i < world; i++) if (dev[i] != dev[0]) other = i;
      if ((rc = sonic_msm_lane_new_on(dev[0], &lane))) return fail("sonic_msm_lane_new_on", rc);
      if ((rc = sonic_dev_alloc_on(dev[0], 64, &dsc))) return fail("sonic_dev_alloc_on", rc);
      rc = sonic_msm_submit(lane, srs[other], 0, 0, dsc, 2);
      if (rc != SONIC_ERR_INVALID_ARG) { fprintf(stderr, "multi_harness: a lane on device %d took an SRS on device %d (status %d)\n", dev[0], dev[other], rc); return 1; }
      sonic_dev_free(dsc);
      sonic_msm_lane_free(lane);
    }
  }

  for (i = 0; i < world; i++) sonic_prover_free(prv[i]);
  for (i = 0; i < world; i++) sonic_srs_free(srs[i]);
  printf("multi_harness: OK (n = %ld, Q = %ld, d = %ld: one proof over %d handles on %s, %ld proofs as a batch, one MSM by term and by bucket range, through the C ABI from one process)\n",
         (long)n, (long)Q, (long)d, world, distinct ? "several GPUs" : "one GPU", (long)K);
  return 0;
}
