/* Plain C99 consumer of libpclseg.so: what a non-Python host of the drop-in boundary links
 * against (include/pclseg.h only — no torch, no C++).  Built and run by tests/test_c_abi.py.
 *
 *   consumer plan       host-only checks (no GPU): version, graph figures of SURVEY.md §8(d),
 *                       status codes and messages for bad descriptors
 *   consumer forward    MI355X: create -> set every weight -> finalize -> forward_raw on host
 *                       buffers; checks the masking contract and softmax normalisation
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pclseg.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
  fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

static pclseg_desc kitti_desc(int h, int w) {
  /* configs/SqueezeSegV2Kitti.py: NUM_CLASS 20, CLASSES[0] == "None", INPUT_MEAN / INPUT_STD */
  static const double mean[5] = {10.88, 0.23, -1.04, 0.21, 12.12};
  static const double std[5] = {11.47, 6.91, 0.86, 0.16, 12.32};
  pclseg_desc d;
  memset(&d, 0, sizeof d);
  d.arch = PCLSEG_ARCH_SQUEEZESEGV2;
  d.height = h; d.width = w; d.num_class = 20; d.none_index = 0; d.output_stride = 16;
  memcpy(d.mean, mean, sizeof mean);
  memcpy(d.std, std, sizeof std);
  return d;
}

static int run_plan(void) {
  pclseg_plan_info info;
  pclseg_desc d = kitti_desc(64, 2048);
  CHECK(pclseg_version() == PCLSEG_VERSION, "library / header version mismatch");
  CHECK(pclseg_plan(&d, &info) == PCLSEG_OK, "%s", pclseg_last_error(NULL));
  CHECK(info.num_params == 937080, "num_params %lld", (long long)info.num_params);            /* SURVEY.md §3.2 */
  CHECK(info.alg_bytes_per_scan == 728367104LL, "alg bytes %lld", (long long)info.alg_bytes_per_scan);
  CHECK(2 * info.alg_macs_per_scan == 26088570880LL, "alg flops %lld", (long long)(2 * info.alg_macs_per_scan));
  d.width = 2040;                                     /* not divisible by 16 */
  CHECK(pclseg_plan(&d, &info) == PCLSEG_ERR_BAD_SHAPE, "W %% 16 must be rejected");
  CHECK(strlen(pclseg_last_error(NULL)) > 0, "no message for the bad shape");
  d = kitti_desc(64, 2048);
  d.arch = 7;
  CHECK(pclseg_plan(&d, &info) == PCLSEG_ERR_BAD_ARG, "unknown arch must be rejected");
  CHECK(pclseg_plan(NULL, &info) == PCLSEG_ERR_BAD_ARG, "NULL desc must be rejected");
  printf("plan ok: %d launches, %lld params, %lld alg bytes/scan\n", info.num_ops,
         937080LL, 728367104LL);
  return 0;
}

static int run_forward(void) {
  enum { H = 32, W = 64, N = 3, NC = 20 };
  pclseg_desc d = kitti_desc(H, W);
  pclseg_handle* h = NULL;
  int i, nw, rc;
  CHECK(pclseg_create(&d, &h) == PCLSEG_OK, "%s", pclseg_last_error(NULL));
  /* forward before finalize is a state error, not a crash */
  {
    float s[5] = {0};
    int32_t p[1];
    CHECK(pclseg_forward_raw(h, s, 1, p, NULL, NULL, NULL, PCLSEG_MEM_HOST) == PCLSEG_ERR_STATE, "state check");
  }
  nw = pclseg_num_weights(h);
  CHECK(nw == 274, "weight tensors %d", nw);
  srand(1);
  for (i = 0; i < nw; ++i) {
    char name[256];
    int64_t shape[4];
    int ndim, k;
    size_t count = 1, j;
    float* w;
    const char* leaf;
    CHECK(pclseg_weight_info(h, i, name, sizeof name, shape, &ndim) == PCLSEG_OK, "%s", pclseg_last_error(h));
    for (k = 0; k < ndim; ++k) count *= (size_t)shape[k];
    w = (float*)malloc(count * sizeof(float));
    leaf = strrchr(name, '/') + 1;
    for (j = 0; j < count; ++j) {
      const float u = (float)rand() / (float)RAND_MAX - 0.5f;
      if (!strcmp(leaf, "kernel")) w[j] = 0.2f * u;
      else if (!strcmp(leaf, "gamma") || !strcmp(leaf, "moving_variance")) w[j] = 1.0f + 0.2f * u;
      else w[j] = 0.1f * u;
    }
    rc = pclseg_set_weight(h, name, w, shape, ndim);
    free(w);
    CHECK(rc == PCLSEG_OK, "%s: %s", name, pclseg_last_error(h));
  }
  {
    const int64_t bad[1] = {3};
    float z[3] = {0};
    CHECK(pclseg_set_weight(h, "conv1/bias", z, bad, 1) == PCLSEG_ERR_BAD_SHAPE, "shape check");
    CHECK(pclseg_set_weight(h, "no/such/tensor", z, bad, 1) == PCLSEG_ERR_MISSING_WEIGHT, "name check");
  }
  CHECK(pclseg_finalize(h) == PCLSEG_OK, "%s", pclseg_last_error(h));
  {
    const size_t px = (size_t)N * H * W;
    float* scans = (float*)calloc(px * 5, sizeof(float));
    int32_t* preds = (int32_t*)malloc(px * sizeof(int32_t));
    float* probs = (float*)malloc(px * NC * sizeof(float));
    uint8_t* mask = (uint8_t*)malloc(px);
    size_t j;
    int c, valid = 0;
    for (j = 0; j < px; ++j) {
      if (j % 3 == 0) continue;                       /* every third pixel has no return: depth 0 */
      for (c = 0; c < 5; ++c) scans[j * 5 + c] = (float)(d.mean[c] + d.std[c] * ((double)rand() / RAND_MAX - 0.5));
      scans[j * 5 + 4] = fabsf(scans[j * 5 + 4]) + 0.05f;
    }
    CHECK(pclseg_forward_raw(h, scans, N, preds, probs, NULL, mask, PCLSEG_MEM_HOST) == PCLSEG_OK, "%s",
          pclseg_last_error(h));
    for (j = 0; j < px; ++j) {
      double sum = 0;
      int best = 0;
      CHECK(mask[j] == (j % 3 != 0), "mask at %lu", (unsigned long)j);
      for (c = 0; c < NC; ++c) {
        sum += probs[j * NC + c];
        if (probs[j * NC + c] > probs[j * NC + best]) best = c;
      }
      CHECK(fabs(sum - 1.0) < 1e-4, "softmax sum %g at %lu", sum, (unsigned long)j);
      CHECK(preds[j] == (mask[j] ? best : d.none_index), "prediction at %lu", (unsigned long)j);
      valid += mask[j];
    }
    printf("forward ok: %d scans, %d valid pixels, argmax/mask contract holds\n", N, valid);
    free(scans); free(preds); free(probs); free(mask);
  }
  CHECK(pclseg_destroy(h) == PCLSEG_OK, "destroy");
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 2 && !strcmp(argv[1], "plan")) return run_plan();
  if (argc == 2 && !strcmp(argv[1], "forward")) return run_forward();
  fprintf(stderr, "usage: consumer plan|forward\n");
  return 2;
}
