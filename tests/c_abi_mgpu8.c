/* c_abi_mgpu8.c -- the N-rank block-cyclic sweep of BASELINE.json config 4 driven through the plain C ABI with VIRTUAL ranks.
 *
 *   c_abi_mgpu8 n d n_star G:flags [G:flags ...]
 *
 * One process, one MI355X: gprc_mgpu_create with the device listed G times gives G ranks with their own buffers and streams that
 * exchange panels by device copies.  The whole protocol of the 8-GPU run executes -- 128 panels dealt to 8 owners at n = 65536,
 * look-ahead chains on the side streams, batched far updates, event recycling, the replicated vector solves, the sliced predict
 * -- and alpha, logp, mean and variance must equal gprc_gpr_fit / gprc_gpr_predict (R/GPRclass.R:127-170) BIT FOR BIT.
 * flags: GPRC_MGPU_NO_LOOKAHEAD (2), GPRC_MGPU_SCATTER_ALLGATHER (4), GPRC_MGPU_AUTO_EXCHANGE (8).
 * Prints one JSON line per variant (gprc_mgpu_stats: a schedule rehearsal, not a scaling measurement) and exits non-zero on any
 * difference.  Inputs follow SURVEY 8(d): X ~ U[-1,1]^d, y = 0.1 sum x^3 + noise, sqrexp l = 1, model noise 0.1; the test points
 * are uniform draws (the bench's grid needs nothing this client checks).  No Python, no torch, no oracle.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gprc_native.h"

static double now_ms(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

static unsigned long long st = 20261004ULL;
static double u01(void) {
  st ^= st << 13; st ^= st >> 7; st ^= st << 17;
  return (double)(st >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s n d n_star G:flags [G:flags ...]\n", argv[0]); return 2; }
  const int64_t n = atoll(argv[1]), d = atoll(argv[2]), ns = atoll(argv[3]);
  if (n < 1 || d < 1 || ns < 1) { fprintf(stderr, "bad sizes\n"); return 2; }
  double* X = (double*)malloc(sizeof(double) * (size_t)(d * n));
  double* y = (double*)malloc(sizeof(double) * (size_t)n);
  double* Xs = (double*)malloc(sizeof(double) * (size_t)(d * ns));
  double *a_ref = (double*)malloc(sizeof(double) * (size_t)n), *a_got = (double*)malloc(sizeof(double) * (size_t)n);
  double *m_ref = (double*)malloc(sizeof(double) * (size_t)ns), *v_ref = (double*)malloc(sizeof(double) * (size_t)ns);
  double *m_got = (double*)malloc(sizeof(double) * (size_t)ns), *v_got = (double*)malloc(sizeof(double) * (size_t)ns);
  if (!X || !y || !Xs || !a_ref || !a_got || !m_ref || !v_ref || !m_got || !v_got) { fprintf(stderr, "out of host memory\n"); return 2; }
  for (int64_t i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int64_t r = 0; r < d; ++r) { const double v = 2.0 * u01() - 1.0; X[i * d + r] = v; acc += v * v * v; }
    y[i] = 0.1 * acc + 0.2 * (u01() - 0.5);
  }
  for (int64_t i = 0; i < d * ns; ++i) Xs[i] = 2.0 * u01() - 1.0;
  const double l = 1.0, noise = 0.1;
  int fails = 0;

  /* the single-rank result: gprc_gpr_fit + gprc_gpr_predict on one context */
  gprc_ctx* ctx = NULL;
  int rc = gprc_ctx_create(0, NULL, &ctx);
  if (rc != 0) { fprintf(stderr, "ctx_create rc=%d (%s)\n", rc, gprc_last_error()); return 1; }
  gprc_model* ref = NULL;
  double logp_ref = 0.0, t0 = now_ms();
  rc = gprc_gpr_fit(ctx, GPRC_SQREXP, &l, 1, X, d, n, y, noise, &ref);
  if (rc != 0) { fprintf(stderr, "reference fit rc=%d (%s)\n", rc, gprc_last_error()); return 1; }
  const double ref_fit_ms = now_ms() - t0;
  t0 = now_ms();
  rc = gprc_gpr_predict(ref, Xs, ns, 1, m_ref, v_ref);
  if (rc != 0) { fprintf(stderr, "reference predict rc=%d (%s)\n", rc, gprc_last_error()); return 1; }
  const double ref_predict_ms = now_ms() - t0;
  if (gprc_gpr_get_alpha(ref, a_ref) != 0 || gprc_gpr_get_logp(ref, &logp_ref) != 0) { fprintf(stderr, "reference getters\n"); return 1; }
  gprc_model_free(ref);      /* 17 GB of factor + the 40 GiB predict chunk at n = 65536: make room for the G replicas */
  gprc_ctx_trim(ctx);
  printf("{\"variant\": \"single rank (first call: allocations included)\", \"n\": %lld, \"d\": %lld, \"n_star\": %lld, \"fit_ms\": %.1f, \"predict_ms\": %.1f}\n",
         (long long)n, (long long)d, (long long)ns, ref_fit_ms, ref_predict_ms);
  fflush(stdout);

  for (int a = 4; a < argc; ++a) {
    int G = 0, flags = 0;
    if (sscanf(argv[a], "%d:%d", &G, &flags) != 2 || G < 1 || G > 64) { fprintf(stderr, "bad variant %s\n", argv[a]); return 2; }
    int devs[64];
    for (int i = 0; i < G; ++i) devs[i] = 0;
    gprc_mgpu* mg = NULL;
    gprc_mgpu_model* mm = NULL;
    rc = gprc_mgpu_create(devs, G, flags, &mg);
    if (rc != 0) { fprintf(stderr, "mgpu_create %s rc=%d (%s)\n", argv[a], rc, gprc_last_error()); return 1; }
    double nu = 0.0, logp = 0.0;
    int attempts = 0;
    rc = gprc_mgpu_gpr_fit_retry(mg, GPRC_SQREXP, &l, 1, X, d, n, y, noise, &mm, &nu, &attempts);
    if (rc != 0) { fprintf(stderr, "mgpu fit %s rc=%d (%s)\n", argv[a], rc, gprc_last_error()); return 1; }
    rc = gprc_mgpu_gpr_predict(mm, Xs, ns, m_got, v_got);
    if (rc != 0) { fprintf(stderr, "mgpu predict %s rc=%d (%s)\n", argv[a], rc, gprc_last_error()); return 1; }
    if (gprc_mgpu_gpr_get_alpha(mm, a_got) != 0 || gprc_mgpu_gpr_get_logp(mm, &logp) != 0) { fprintf(stderr, "mgpu getters\n"); return 1; }
    const int alpha_ok = memcmp(a_got, a_ref, sizeof(double) * (size_t)n) == 0;
    const int logp_ok = logp == logp_ref && attempts == 1 && nu == noise;
    const int mean_ok = memcmp(m_got, m_ref, sizeof(double) * (size_t)ns) == 0;
    const int var_ok = memcmp(v_got, v_ref, sizeof(double) * (size_t)ns) == 0;
    /* every rank holds the whole factor: the LAST rank's replica predicts a few points alone */
    gprc_model* last = NULL;
    const int64_t few = ns < 256 ? ns : 256;
    int replica_ok = gprc_mgpu_model_rank(mm, G - 1, &last) == 0 && last &&
                     gprc_gpr_predict(last, Xs, few, 1, m_got, v_got) == 0 &&
                     memcmp(m_got, m_ref, sizeof(double) * (size_t)few) == 0 && memcmp(v_got, v_ref, sizeof(double) * (size_t)few) == 0;
    double stt[10 + 3 * 64];
    memset(stt, 0, sizeof stt);
    if (gprc_mgpu_stats(mg, stt, 10 + 3 * G) != 0) { fprintf(stderr, "mgpu_stats (%s)\n", gprc_last_error()); return 1; }
    printf("{\"variant\": \"%s\", \"ranks\": %d, \"flags\": %d, \"panels\": %.0f, \"exchange_mode\": %.0f, \"exchange_ops\": %.0f, "
           "\"gb_in_per_rank\": %.3f, \"event_pairs\": %.0f, \"far_passes\": %.0f, \"lookahead_updates\": %.0f, \"fit_ms\": %.1f, "
           "\"predict_ms\": %.1f, \"bitwise\": {\"alpha\": %s, \"logp\": %s, \"mean\": %s, \"var\": %s, \"last_rank_replica\": %s}, \"per_rank\": [",
           argv[a], G, flags, stt[1], stt[2], stt[3], stt[4] / 1e9, stt[5], stt[6], stt[7], stt[8], stt[9], alpha_ok ? "true" : "false",
           logp_ok ? "true" : "false", mean_ok ? "true" : "false", var_ok ? "true" : "false", replica_ok ? "true" : "false");
    for (int r = 0; r < G; ++r)
      printf("%s{\"rank\": %d, \"fill_sweep_ms\": %.1f, \"alpha_logp_ms\": %.1f, \"predict_ms\": %.1f}", r ? ", " : "", r, stt[10 + 3 * r],
             stt[11 + 3 * r], stt[12 + 3 * r]);
    printf("]}\n");
    fflush(stdout);
    if (!(alpha_ok && logp_ok && mean_ok && var_ok && replica_ok)) ++fails;
    gprc_mgpu_model_free(mm);
    gprc_mgpu_destroy(mg);
  }
  gprc_ctx_destroy(ctx);
  printf(fails ? "c_abi_mgpu8: %d variant(s) DIFFER\n" : "c_abi_mgpu8: all variants bit-identical\n", fails);
  free(X); free(y); free(Xs); free(a_ref); free(a_got); free(m_ref); free(v_ref); free(m_got); free(v_got);
  return fails ? 1 : 0;
}
