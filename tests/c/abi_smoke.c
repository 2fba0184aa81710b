/* A plain-C client of libbogp.so: proves that include/bogp.h is a C header (no C++, no torch types) and that the
 * library is usable without Python.  Built by tests (gcc -std=c99); run on a GPU box by the -m gpu test, which
 * compares the printed numbers with the same calls made through ctypes.
 *
 *   abi_smoke N d M seed   -> prints "llf <v>", "best <value> <index>", "mu0 <v>", "mse0 <v>"                      */
#include <stdio.h>
#include <stdlib.h>

#include "bogp.h"

static double lcg(unsigned long long* s) { /* any deterministic numbers will do; the test regenerates them the same way */
  *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(*s >> 11) / 9007199254740992.0;
}

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ != BOGP_OK) {                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bogp_last_error(h));     \
      return 2;                                                              \
    }                                                                        \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 5) return 1;
  const int N = atoi(argv[1]), d = atoi(argv[2]);
  const long long M = atoll(argv[3]);
  unsigned long long s = strtoull(argv[4], NULL, 10);
  double* X = (double*)malloc(sizeof(double) * N * d);
  double* y = (double*)malloc(sizeof(double) * N);
  double* par = (double*)malloc(sizeof(double) * (d + 1));
  double* lo = (double*)malloc(sizeof(double) * d);
  double* hi = (double*)malloc(sizeof(double) * d);
  int i, k;
  for (i = 0; i < N; ++i) {
    double acc = 0.0;
    for (k = 0; k < d; ++k) {
      X[i * d + k] = -5.0 + 10.0 * lcg(&s);
      acc += X[i * d + k] * X[i * d + k];
    }
    y[i] = acc / (8.0 * d) - 1.0;
  }
  for (k = 0; k < d; ++k) { par[k] = 0.02; lo[k] = -5.0; hi[k] = 5.0; }
  par[d] = 0.9;

  bogp_handle* h = NULL;
  if (bogp_create(0, &h) != BOGP_OK) {
    fprintf(stderr, "bogp_create: %s\n", bogp_last_error(NULL));
    return 3;
  }
  if (bogp_abi_version() != BOGP_ABI_VERSION) return 4;
  double llf = 0.0;
  CHECK(bogp_set_train(h, X, y, N, d, 1));
  CHECK(bogp_commit(h, BOGP_KERNEL_MATERN32, BOGP_MODE_NOISY, par, d + 1, 1e-6, BOGP_TREND_CONSTANT, 0, 0.0, &llf));
  CHECK(bogp_candidates_generate(h, lo, hi, M, 42ULL, 0));
  {
    const int acq_id[1] = {BOGP_ACQ_EI};
    const double acq_par[1] = {0.0};
    double ymin = y[0], best = 0.0;
    int64_t idx = -1;
    double* mu = (double*)malloc(sizeof(double) * M);
    double* mse = (double*)malloc(sizeof(double) * M);
    for (i = 1; i < N; ++i) if (y[i] < ymin) ymin = y[i];
    CHECK(bogp_sweep(h, 1, acq_id, acq_par, ymin, 1, &best, &idx, NULL));
    CHECK(bogp_predict(h, mu, mse));
    printf("llf %.17g\nbest %.17g %lld\nmu0 %.17g\nmse0 %.17g\n", llf, best, (long long)idx, mu[0], mse[0]);
    { /* a plain-C client shards too: the library's own exchange (one-rank communicator here), global index = local + 1000 */
      unsigned char id[BOGP_COMM_ID_BYTES];
      double gbest = 0.0;
      int64_t gidx = -1;
      int rank = -1, world = -1;
      double* xb = (double*)malloc(sizeof(double) * d);
      double* xr = (double*)malloc(sizeof(double) * d);
      if (bogp_comm_unique_id(id) != BOGP_OK) return 5;
      CHECK(bogp_comm_init(h, id, 0, 1));
      CHECK(bogp_comm_info(h, &rank, &world));
      CHECK(bogp_sweep(h, 1, acq_id, acq_par, ymin, 1, &best, &idx, NULL));
      CHECK(bogp_exchange_argmax(h, 1000, &gbest, &gidx, xb));
      CHECK(bogp_candidates_read(h, &idx, 1, xr));
      printf("exchange %.17g %lld %d %d %.17g %.17g\n", gbest, (long long)gidx, rank, world, xb[d - 1], xr[d - 1]);
      free(xb);
      free(xr);
    }
    { /* the consumption path of the reference's inner optimisers through the same boundary (r03): one point with the moments
         and the criterion's input-gradient; then the sweep's winner polished on the device */
      double* x0 = (double*)malloc(sizeof(double) * d);
      double* xo = (double*)malloc(sizeof(double) * d);
      double* dmu = (double*)malloc(sizeof(double) * d);
      double* dms = (double*)malloc(sizeof(double) * d);
      double* dei = (double*)malloc(sizeof(double) * d);
      double pmu = 0.0, pms = 0.0, pei = 0.0, fo = 0.0;
      int ne = 0;
      CHECK(bogp_candidates_read(h, &idx, 1, x0));
      CHECK(bogp_point_eval_batch(h, x0, 1, 1, acq_id, acq_par, ymin, 1, &pmu, &pms, dmu, dms, &pei, dei));
      CHECK(bogp_polish(h, x0, 1, lo, hi, BOGP_ACQ_EI, 0.0, ymin, 1, 50, 1e-8, 1e6, xo, &fo, &ne));
      printf("point %.17g %.17g %.17g %.17g %.17g\npolish %.17g %d %.17g\n", pmu, pms, pei, dmu[0], dei[d - 1], fo, ne, xo[0]);
      free(x0); free(xo); free(dmu); free(dms); free(dei);
    }
    free(mu);
    free(mse);
  }
  bogp_destroy(h);
  free(X); free(y); free(par); free(lo); free(hi);
  return 0;
}
