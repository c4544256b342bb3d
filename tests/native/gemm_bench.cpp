// GEMM micro-benchmark (test/bench infrastructure): times the shapes of the Q-Former path with both
// main loops interleaved in one process (A/B as cdna_hip_programming.md 5.4 rule 24 asks) on random
// data.  Usage: gemm_bench [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels.h"

using namespace mra;

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(99);                                                                              \
    }                                                                                        \
  } while (0)

struct Shape {
  const char* name;
  int M, N, K, epi, cfg;
};

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const int only = argc > 2 ? atoi(argv[2]) : -1;      // run a single shape (profiling)
  const int only_variant = argc > 3 ? atoi(argv[3]) : -1;
  const Shape shapes[] = {
      {"kvproj video 32x8224 (headline)", 32 * 8224, 9216, 1408, EPI_KV, -1},
      {"kvproj video 32x257 (ref)", 32 * 257, 9216, 1408, EPI_KV, -1},
      {"kvproj audio 32x496", 32 * 496, 9216, 768, EPI_KV, -1},
      {"qkv 2048x768->2304", 2048, 2304, 768, EPI_OP, -1},
      {"qkv 2048x768->2304 cfg1", 2048, 2304, 768, EPI_OP, 1},
      {"attn-out 2048x768->768", 2048, 768, 768, EPI_RES_F32, -1},
      {"cross-q 1024x768->768", 1024, 768, 768, EPI_OP, -1},
      {"ffn-up 2048x768->3072", 2048, 3072, 768, EPI_GELU_OP, -1},
      {"ffn-up 2048x768->3072 cfg0", 2048, 3072, 768, EPI_GELU_OP, 0},
      {"ffn-down 2048x3072->768", 2048, 768, 3072, EPI_RES_F32, -1},
      {"ffn-down 2048x3072->768 cfg1", 2048, 768, 3072, EPI_RES_F32, 1},
      // proxies for the folded cross-attention: scores enc . Q'^T and P . enc per layer, all 32 items
      {"fold scores N384 128-tile", 32 * 8224, 384, 1408, EPI_OP, 1},
      {"fold scores N512 256-tile", 32 * 8224, 512, 1408, EPI_OP, 2},
      {"fold scores N512 f32 out", 32 * 8224, 512, 1408, EPI_F32, 2},
      {"fold P.enc M12288 N1408 K8256 128", 32 * 384, 1408, 8256, EPI_OP, 1},
      {"fold P.enc M16384 N1536 K8256 256", 32 * 512, 1536, 8256, EPI_OP, 2},
  };
  size_t maxA = 0, maxW = 0, maxC = 0;
  for (auto& s : shapes) {
    maxA = std::max(maxA, (size_t)s.M * s.K);
    maxW = std::max(maxW, (size_t)s.N * s.K);
    maxC = std::max(maxC, (size_t)s.M * s.N);
  }
  // random f16 bit patterns in [-2, 2): sign 1 bit, exponent 01111/10000-ish -> build from floats on the host
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  auto fill = [&](size_t n) {
    std::vector<_Float16> h(std::min(n, (size_t)1 << 24));
    for (auto& v : h) v = (_Float16)d(rng);
    _Float16* p;
    CK(hipMalloc((void**)&p, n * 2));
    for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(p + off, h.data(), std::min(h.size(), n - off) * 2, hipMemcpyHostToDevice));
    return p;
  };
  _Float16* A = fill(maxA);
  _Float16* W = fill(maxW);
  float *bias, *R, *C;
  CK(hipMalloc((void**)&bias, 16384 * 4));
  CK(hipMemset(bias, 0, 16384 * 4));
  CK(hipMalloc((void**)&R, (size_t)2048 * 768 * 4));
  CK(hipMemset(R, 0, (size_t)2048 * 768 * 4));  // only the EPI_RES_F32 shapes (M <= 2048, N = 768) read it
  CK(hipMalloc((void**)&C, maxC * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-36s %10s %10s %10s %10s %10s %8s\n", "shape", "ring TF/s", "v1 TF/s", "v1+pf TF/s", "spread", "ws(256)", "v1 us");
  int shape_idx = -1;
  for (auto& s : shapes) {
    ++shape_idx;
    if (only >= 0 && shape_idx != only) continue;
    GemmProb p{};
    p.A = A; p.a = RowView{0, s.M, s.K};
    p.W = W; p.bias = bias;
    p.C = C; p.c = RowView{0, s.M, s.N};
    p.R = R; p.r = RowView{0, s.M, s.N};
    p.M = s.M; p.N = s.N; p.K = s.K;
    if (s.epi == EPI_KV) { p.kv_tokens = s.M / 32; p.kv_items = 32; p.kv_heads = 12; }
    gemm_force_config(s.cfg);
    double best[10] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
    const int reps = s.M > 100000 ? 3 : 20;
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < 10; ++v) {
        if (v == 4 || (v == 6 && s.epi != EPI_KV)) continue;
        if (only_variant >= 0 && v != only_variant) continue;
        gemm_force_variant(v);
        if (launch_gemm(&p, 1, s.epi, OP_F16, 0)) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch_gemm(&p, 1, s.epi, OP_F16, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best[v] = std::min(best[v], (double)ms / reps);
      }
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("%-36s %10.1f %10.1f %10.1f %10.1f %10.1f %8.1f\n", s.name, fl / best[0] / 1e9, fl / best[1] / 1e9, fl / best[2] / 1e9, fl / best[3] / 1e9, fl / best[5] / 1e9, best[1] * 1e3);
    printf("    ws2 (flag hand-off, no K-loop barrier): %.1f TF/s   k128 small tiles: %.1f TF/s (%.1f us)   rot: %.1f TF/s\n", fl / best[7] / 1e9, fl / best[8] / 1e9, best[8] * 1e3, fl / best[9] / 1e9);
    if (s.epi == EPI_KV) printf("    (diagnostic, wrong results) ws without DMA after tile 1: %.1f TF/s\n", fl / best[6] / 1e9);
    fflush(stdout);
  }
  if (only == 0) {  // stamped diagnostic build of the two-buffer loop on the headline shape
    unsigned long long* dbg;
    const size_t nent = (size_t)4096 * 8 * 8;
    CK(hipMalloc((void**)&dbg, nent * 8));
    CK(hipMemset(dbg, 0, nent * 8));
    gemm_set_debug_buffer(dbg);
    gemm_force_variant(4);
    gemm_force_config(-1);
    const Shape& s = shapes[0];
    GemmProb p{};
    p.A = A; p.a = RowView{0, s.M, s.K};
    p.W = W; p.bias = bias; p.C = C; p.c = RowView{0, s.M, s.N};
    p.M = s.M; p.N = s.N; p.K = s.K; p.kv_tokens = s.M / 32; p.kv_items = 32; p.kv_heads = 12;
    launch_gemm(&p, 1, EPI_KV, OP_F16, 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nent);
    CK(hipMemcpy(h.data(), dbg, nent * 8, hipMemcpyDeviceToHost));
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    size_t n = 0;
    for (size_t i = 0; i < nent; i += 8)
      if (h[i] | h[i + 3]) { for (int e = 0; e < 7; ++e) sum[e] += (double)h[i + e]; ++n; }
    const double iters = 22.0;
    printf("stamped v1, per K-tile iteration, mean over %zu waves (shader cycles): vmcnt-wait %.0f  barrier %.0f  dma-issue %.0f  reads+mfma %.0f\n",
           n, sum[0] / n / iters, sum[1] / n / iters, sum[2] / n / iters, sum[3] / n / iters);
    printf("per tile (cycles): prologue %.0f  k-loop %.0f  epilogue+store drain %.0f\n", sum[4] / n, sum[5] / n, sum[6] / n);
    gemm_set_debug_buffer(nullptr);
  }
  gemm_force_variant(5);
  gemm_force_config(-1);
  return 0;
}
