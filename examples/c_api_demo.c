/*
 * Plain-C host program on the C ABI (include/tristage.h): no Python, no torch.
 *   gcc -O2 -Iinclude examples/c_api_demo.c -o examples/c_api_demo \
 *       -Ltristage-rag_amd -ltristage -Wl,-rpath,$PWD/tristage-rag_amd -lm
 *   ./examples/c_api_demo <n> <d> <k> <nq> <out.bin>
 * Builds a deterministic corpus on the host, adds it (host pointers), searches,
 * and writes  [float scores nq*k][int64 ids nq*k]  so a test can compare with the
 * oracle computed from the same generator (tests/test_c_api_gpu.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "tristage.h"

static uint64_t s_state;
static float next_unit(void) { /* xorshift64* -> (-1, 1) */
  s_state ^= s_state >> 12; s_state ^= s_state << 25; s_state ^= s_state >> 27;
  return (float)((double)((s_state * 2685821657736338717ULL) >> 11) / 4503599627370496.0 - 1.0);
}

static void fill(float* x, int64_t n, int d, uint64_t seed) {
  s_state = seed;
  for (int64_t i = 0; i < n; ++i) {
    double ss = 0.0;
    for (int j = 0; j < d; ++j) { x[i * d + j] = next_unit(); ss += (double)x[i * d + j] * x[i * d + j]; }
    const float den = (float)sqrt(ss) + 1e-8f;
    for (int j = 0; j < d; ++j) x[i * d + j] /= den;
  }
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != TS_OK) { \
  fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ts_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s n d k nq out.bin\n", argv[0]); return 2; }
  const int64_t n = atoll(argv[1]);
  const int d = atoi(argv[2]), k = atoi(argv[3]), nq = atoi(argv[4]);
  float* corpus = (float*)malloc(sizeof(float) * (size_t)n * d);
  float* queries = (float*)malloc(sizeof(float) * (size_t)nq * d);
  float* D = (float*)malloc(sizeof(float) * (size_t)nq * k);
  int64_t* I = (int64_t*)malloc(sizeof(int64_t) * (size_t)nq * k);
  if (!corpus || !queries || !D || !I) return 3;
  fill(corpus, n, d, 0x9E3779B97F4A7C15ULL);
  fill(queries, nq, d, 0xD1B54A32D192ED03ULL);

  ts_index* h = NULL;
  CHECK(ts_index_create(d, TS_F32, TS_METRIC_INNER_PRODUCT, 0, &h));
  if (ts_index_search(h, queries, nq, TS_F32, k, D, I, TS_FLAG_HOST_PTR, NULL) != TS_ERR_EMPTY) {
    fprintf(stderr, "search on an empty index must return TS_ERR_EMPTY\n");
    return 4;
  }
  /* two appends, the second one unaligned to the 32-row tile */
  const int64_t first = n / 3 + 5;
  CHECK(ts_index_add(h, corpus, first, TS_F32, TS_FLAG_HOST_PTR, NULL));
  CHECK(ts_index_add(h, corpus + first * d, n - first, TS_F32, TS_FLAG_HOST_PTR, NULL));
  if (ts_index_ntotal(h) != n || ts_index_dim(h) != d) return 5;
  CHECK(ts_index_search(h, queries, nq, TS_F32, k, D, I, TS_FLAG_HOST_PTR, NULL));
  int64_t info[4];
  CHECK(ts_index_last_search_info(h, info));
  printf("abi %d  ntotal %lld  path %lld  best[0] id %lld score %.6f\n", ts_abi_version(),
         (long long)ts_index_ntotal(h), (long long)info[0], (long long)I[0], D[0]);
  FILE* f = fopen(argv[5], "wb");
  if (!f) return 6;
  fwrite(D, sizeof(float), (size_t)nq * k, f);
  fwrite(I, sizeof(int64_t), (size_t)nq * k, f);
  fwrite(corpus, sizeof(float), (size_t)n * d, f);
  fwrite(queries, sizeof(float), (size_t)nq * d, f);
  fclose(f);
  CHECK(ts_index_destroy(h));
  free(corpus); free(queries); free(D); free(I);
  return 0;
}
