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

#include <cstring>

// A/B of the folded cross-attention's two big products at the headline shape (32 items x Kv 8224 x E 1408, 384 rows):
// streaming kernels (fold_stream.hip) against the loader-wave GEMMs + rescale pass they replaced.  Random data.
static void fold_ab(int rounds) {
  const int items = 32, kv = 8224, E = 1408, R = 384;
  const int kvp = (std::max((kv + 127) / 128 * 128, (kv + 175) / 176 * 176) + 127) / 128 * 128, sld = fold_stream_stat_ld(kvp), nt = (kv + 175) / 176;
  std::mt19937 rng(3);
  std::normal_distribution<float> d(0.f, 1.f);
  std::vector<_Float16> hq((size_t)items * R * E), hx((size_t)1 << 24);
  for (auto& v : hq) v = (_Float16)(d(rng) * 0.5f);
  for (auto& v : hx) v = (_Float16)d(rng);
  _Float16 *Q, *Qb, *X, *P, *U, *G;
  float *M, *L, *I;
  CK(hipMalloc((void**)&Qb, hq.size() * 2));
  CK(hipMalloc((void**)&Q, hq.size() * 2)); CK(hipMemcpy(Q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
  const size_t nx = (size_t)items * kv * E;
  CK(hipMalloc((void**)&X, nx * 2));
  for (size_t off = 0; off < nx; off += hx.size()) CK(hipMemcpy(X + off, hx.data(), std::min(hx.size(), nx - off) * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&P, (size_t)items * R * kvp * 2)); CK(hipMalloc((void**)&U, (size_t)items * R * E * 2));
  CK(hipMalloc((void**)&G, (size_t)items * R * sld * 2)); CK(hipMalloc((void**)&M, (size_t)items * R * sld * 4));
  CK(hipMalloc((void**)&L, (size_t)items * R * sld * 4)); CK(hipMalloc((void**)&I, (size_t)items * R * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto&& fn) {
    double best = 1e30;
    for (int r = 0; r < rounds; ++r) {
      fn();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) fn();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, (double)ms / 5);
    }
    return best;
  };
  FoldStreamArgs a{};
  a.qp = Q; a.qpb = Qb; a.enc = X; a.p = P; a.u = U; a.stat_m = M; a.stat_l = L; a.gexp = G; a.ginv = I;
  a.items = items; a.kv = kv; a.kvp = kvp; a.E = E; a.alpha = 0.125f * 1.4426950408889634f;
  a.phase = 3; if (launch_fold_stream(a, 0)) { printf("fold stream launch failed\n"); return; }
  CK(hipDeviceSynchronize());
  a.phase = 1; const double t_s = timed([&] { launch_fold_stream(a, 0); });
  a.phase = 2; const double t_p = timed([&] { launch_fold_stream(a, 0); });
  a.phase = 3; const double t_b = timed([&] { launch_fold_stream(a, 0); });
  // the loader-wave formulation
  GemmProb sc{};
  sc.A = Q; sc.a = RowView{0, R, E}; sc.a_bs = (long long)R * E; sc.W = X; sc.w_bs = (long long)kv * E;
  sc.M = R; sc.N = kv; sc.K = E; sc.batch = items; sc.n_ragged = 1; sc.tile_cfg = 5;
  sc.C = P; sc.c = RowView{0, R, kvp}; sc.c_bs_bytes = (long long)R * kvp * 2; sc.alpha = a.alpha; sc.stat_m = M; sc.stat_l = L;
  GemmProb pv{};
  pv.A = P; pv.a = RowView{0, R, kvp}; pv.a_bs = (long long)R * kvp; pv.W = X; pv.w_bs = (long long)kv * E; pv.w_ld = E; pv.k_rows = kv;
  pv.C = U; pv.c = RowView{0, R, E}; pv.c_bs_bytes = (long long)R * E * 2; pv.M = R; pv.N = E; pv.K = kvp; pv.batch = items; pv.tile_cfg = 5;
  gemm_force_config(-1); gemm_force_variant(5);
  const double o_s = timed([&] { launch_gemm(&sc, 1, EPI_SOFTPART, OP_F16, 0); });
  const double o_r = timed([&] { launch_softmax_rescale(P, kvp, M, L, items * R, nt, 176, kvp, OP_F16, 0); });
  const double o_p = timed([&] { launch_gemm(&pv, 1, EPI_OP, OP_F16, 0); });
  // the rescale pass folded into P . enc: row factors from a small kernel, applied to the P~ fragments in registers
  float* F;
  CK(hipMalloc((void**)&F, (size_t)items * nt * 512 * 4));
  const double o_f = timed([&] { launch_fold_rowfactor(M, L, F, items * R, R, nt, P, kvp, 176, kvp, 0); });
  pv.pscale = F; pv.ps_ntiles = nt;
  const double o_ps = timed([&] { launch_gemm(&pv, 1, EPI_OP, OP_F16, 0); });
  printf("in-register rescale: row factors %.3f ms + pv %.3f ms (against rescale pass %.3f + pv %.3f)\n", o_f, o_ps, o_r, o_p);
  const double fl = 2.0 * items * R * (double)kv * E;
  printf("fold A/B (ms, executed TF/s): streaming scores+stats %.3f (%.0f)  pv %.3f (%.0f)  both %.3f | loader-wave scores %.3f (%.0f)  rescale %.3f  pv %.3f (%.0f)  sum %.3f\n",
         t_s, fl / t_s / 1e9, t_p, fl / t_p / 1e9, t_b, o_s, fl / o_s / 1e9, o_r, o_p, fl / o_p / 1e9, o_s + o_r + o_p);
}

// tile orders of the 256 x 256 loader-wave kernel on the big shapes (K/V projection, the ViT's four GEMMs over 1024 frames)
static void order_ab(int rounds) {
  const Shape shapes[] = {
      {"kvproj video 32x8224", 32 * 8224, 9216, 1408, EPI_KV, -1},
      {"vit qkv 263168x1408->4608", 1024 * 257, 4608, 1408, EPI_OP, -1},
      {"vit fc1 263168x1408->6144", 1024 * 257, 6144, 1408, EPI_GELU_OP, -1},
      {"  same shape, plain epilogue", 1024 * 257, 6144, 1408, EPI_OP, -1},
      {"vit fc2 263168x6144->1408", 1024 * 257, 1408, 6144, EPI_F32, 3},
      {"vit proj 263168x1408->1408", 1024 * 257, 1408, 1408, EPI_F32, 3},
  };
  const size_t nA = (size_t)1024 * 257 * 6144, nW = (size_t)9216 * 1408, nC = (size_t)1024 * 257 * 9216;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)1 << 24);
  for (auto& v : h) v = (_Float16)d(rng);
  _Float16 *A, *W, *C;
  float* bias;
  CK(hipMalloc((void**)&A, nA * 2)); CK(hipMalloc((void**)&W, nW * 2)); CK(hipMalloc((void**)&C, nC * 2));
  for (size_t off = 0; off < nA; off += h.size()) CK(hipMemcpy(A + off, h.data(), std::min(h.size(), nA - off) * 2, hipMemcpyHostToDevice));
  for (size_t off = 0; off < nW; off += h.size()) CK(hipMemcpy(W + off, h.data(), std::min(h.size(), nW - off) * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&bias, 16384 * 4)); CK(hipMemset(bias, 0, 16384 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int orders[] = {0x10000, 0x20000, 0x10000, 0x20000, 0x20008};   // 0x20000: the eight-phase kernel   // 0x10000: the default walk (rows fastest in panels of 8 row tiles)
  for (auto& s : shapes) {
    GemmProb p{};
    p.A = A; p.a = RowView{0, s.M, s.K}; p.W = W; p.bias = bias; p.C = C; p.c = RowView{0, s.M, s.N};
    p.M = s.M; p.N = s.N; p.K = s.K; p.tile_cfg = s.cfg > 0 ? s.cfg : 0; p.n_mask = s.N % 256 != 0;
    if (s.epi == EPI_KV) { p.kv_tokens = s.M / 32; p.kv_items = 32; p.kv_heads = 12; }
    printf("%-30s", s.name);
    for (int o : orders) {
      gemm_set_tile_order(o & 0x1ffff ? (o & 0x1ffff) : 0x10000);
      gemm_set_eight_phase(o & 0x20000 ? 1 : 0);
      double best = 1e30;
      for (int r = 0; r < rounds; ++r) {
        if (launch_gemm(&p, 1, s.epi, OP_F16, 0)) { printf("launch failed\n"); return; }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 3; ++i) launch_gemm(&p, 1, s.epi, OP_F16, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms / 3);
      }
      printf("  %s gn%-2d %5.0f", o & 0x20000 ? "p8" : "ws", o & 0xff, 2.0 * s.M * s.N * s.K / best / 1e9);
    }
    printf("  TF/s\n");
    fflush(stdout);
  }
  gemm_set_tile_order(0);
  gemm_set_eight_phase(1);
}

// would split-K pay on the layer chain's residual GEMMs?  One problem over the whole K against two problems over half of it each
// (a two-problem launch: exactly the work of a 2-way split, partial sums to two fp32 outputs).
static void splitk_ab(int rounds) {
  const int M = 2048, N = 768;
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)M * 3072);
  for (auto& v : h) v = (_Float16)d(rng);
  _Float16 *A, *W;
  float *C, *R, *bias;
  CK(hipMalloc((void**)&A, h.size() * 2)); CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&W, h.size() * 2)); CK(hipMemcpy(W, h.data(), (size_t)N * 3072 * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&C, (size_t)4 * M * N * 4)); CK(hipMalloc((void**)&R, (size_t)M * N * 4)); CK(hipMemset(R, 0, (size_t)M * N * 4));
  CK(hipMalloc((void**)&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto&& fn) {
    double best = 1e30;
    for (int r = 0; r < rounds; ++r) {
      fn();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) fn();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, (double)ms / 20);
    }
    return best * 1e3;
  };
  for (int K : {768, 3072}) {
    for (int rows : {1024, 2048}) {
      GemmProb p{};
      p.A = A; p.a = RowView{0, rows, K}; p.W = W; p.bias = bias; p.R = R; p.r = RowView{0, rows, N}; p.C = C; p.c = RowView{0, rows, N};
      p.M = rows; p.N = N; p.K = K;
      if (K == 3072) p.tile_cfg = 6;
      const double whole = timed([&] { launch_gemm(&p, 1, EPI_RES_F32, OP_F16, 0); });
      // halves: A row stride K, but only K / 2 columns are walked -> RowView.ld = K with p.K = K / 2; W needs its own row stride: use two
      // weight matrices of K / 2 columns (same bytes walked)
      GemmProb q[2];
      for (int s = 0; s < 2; ++s) {
        q[s] = GemmProb{};
        q[s].A = A + s * (K / 2); q[s].a = RowView{0, rows, K}; q[s].W = W + (size_t)s * N * (K / 2); q[s].C = C + (size_t)(1 + s) * rows * N; q[s].c = RowView{0, rows, N};
        q[s].M = rows; q[s].N = N; q[s].K = K / 2;
        if (K == 3072) q[s].tile_cfg = 6;
      }
      const double halves = timed([&] { launch_gemm(q, 2, EPI_F32, OP_F16, 0); });
      q[0].tile_cfg = q[1].tile_cfg = 0;
      const double halves_auto = timed([&] { launch_gemm(q, 2, EPI_F32, OP_F16, 0); });
      printf("rows %d K %d: whole %.1f us   two halves %.1f us (auto tile %.1f us)\n", rows, K, whole, halves, halves_auto);
    }
  }
}

// the layer chain's GEMM shapes on every small-tile configuration (GemmProb::tile_cfg 1 = 64 x 64, 2 = 128 x 128, 6 = 64 x 128 x 128-deep)
static void chain_ab(int rounds) {
  struct S { const char* name; int M, N, K, epi, groups; };
  const S shapes[] = {{"qkv 2048x768->2304", 2048, 2304, 768, EPI_OP, 1},          {"attn-out 2048x768->768", 2048, 768, 768, EPI_RES_F32, 1},
                      {"cross-q 1024x768->768", 1024, 768, 768, EPI_OP, 1},        {"cross-out 1024x768->768", 1024, 768, 768, EPI_RES_F32, 1},
                      {"ffn-up 2 x 1024x768->3072", 1024, 3072, 768, EPI_GELU_OP, 2}, {"ffn-down 2 x 1024x3072->768", 1024, 768, 3072, EPI_RES_F32, 2}};
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)3072 * 3072);
  for (auto& v : h) v = (_Float16)d(rng);
  _Float16 *A, *W;
  float *C, *R, *bias;
  CK(hipMalloc((void**)&A, h.size() * 2)); CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&W, h.size() * 2 * 2)); CK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W + h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&C, (size_t)2 * 2048 * 3072 * 4)); CK(hipMalloc((void**)&R, (size_t)2048 * 768 * 4)); CK(hipMemset(R, 0, (size_t)2048 * 768 * 4));
  CK(hipMalloc((void**)&bias, 4096 * 4)); CK(hipMemset(bias, 0, 4096 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& sh : shapes) {
    printf("%-30s", sh.name);
    for (int cfg : {1, 2, 6, 9, 10, 11}) {
      GemmProb q[2];
      for (int g = 0; g < sh.groups; ++g) {
        q[g] = GemmProb{};
        q[g].A = A + (size_t)g * 1024 * sh.K; q[g].a = RowView{0, sh.M, sh.K}; q[g].W = W + (size_t)g * h.size(); q[g].bias = bias;
        q[g].R = R; q[g].r = RowView{0, sh.M, sh.N}; q[g].C = (char*)C + (size_t)g * 2048 * 3072 * 2; q[g].c = RowView{0, sh.M, sh.N};
        q[g].M = sh.M; q[g].N = sh.N; q[g].K = sh.K; q[g].tile_cfg = cfg;
      }
      if (cfg == 6 && sh.K % 128) { printf("  cfg6   n/a"); continue; }
      if ((cfg == 9 && sh.N % 144) || (cfg == 10 && sh.N % 192) || (cfg == 11 && sh.N % 96)) { printf("  cfg%-2d  n/a   ", cfg); continue; }
      double best = 1e30;
      bool ok = true;
      for (int r = 0; r < rounds && ok; ++r) {
        if (launch_gemm(q, sh.groups, sh.epi, OP_F16, 0)) { ok = false; break; }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch_gemm(q, sh.groups, sh.epi, OP_F16, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms / 20 * 1e3);
      }
      if (ok) printf("  cfg%d %5.1f us", cfg, best); else printf("  cfg%d refused", cfg);
    }
    printf("\n");
  }
}

// stamped timeline of the ring kernel on the QKV shape (2048 x 768 -> 2304, 256 workgroups of 144 x 128): wall clock (10 ns ticks) relative to the
// earliest workgroup start, mean / min / max over the workgroups' wave 0 and wave 4
static void ring_stamp() {
  const int M = 2048, N = 2304, K = 768;
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)N * K);
  for (auto& v : h) v = (_Float16)d(rng);
  _Float16 *A, *W, *C;
  float* bias;
  CK(hipMalloc((void**)&A, h.size() * 2)); CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&W, h.size() * 2)); CK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&C, (size_t)M * N * 2)); CK(hipMalloc((void**)&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
  unsigned long long* dbg;
  const size_t nent = (size_t)256 * 8 * 16;
  CK(hipMalloc((void**)&dbg, nent * 8));
  GemmProb p{};
  p.A = A; p.a = RowView{0, M, K}; p.W = W; p.bias = bias; p.C = C; p.c = RowView{0, M, N}; p.M = M; p.N = N; p.K = K; p.tile_cfg = 9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(dbg, 0, nent * 8));
    gemm_set_debug_buffer(dbg);
    for (int i = 0; i < 3; ++i) launch_gemm(&p, 1, EPI_OP, OP_F16, 0);   // the last launch's stamps stay
    CK(hipDeviceSynchronize());
    gemm_set_debug_buffer(nullptr);
    std::vector<unsigned long long> s(nent);
    CK(hipMemcpy(s.data(), dbg, nent * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 256; ++b) t0 = std::min(t0, s[((size_t)b * 8) * 16]);
    const char* names[8] = {"start", "touches issued", "prologue staged", "barrier 0 passed", "barrier 4 passed", "K loop done", "groups reduced", "end"};
    for (int w : {0, 4}) {
      printf("wave %d (wall clock, us after the first workgroup's start; shader cycles since this wave's start):\n", w);
      for (int e = 0; e < 8; ++e) {
        double sum = 0, mn = 1e30, mx = 0, cyc = 0;
        for (int b = 0; b < 256; ++b) {
          const unsigned long long* q = &s[((size_t)b * 8 + w) * 16];
          const double us = (double)(q[e] - t0) * 0.01;
          sum += us; mn = std::min(mn, us); mx = std::max(mx, us); cyc += (double)(q[8 + e] - q[8]);
        }
        printf("  %-18s mean %6.2f  min %6.2f  max %6.2f   cycles %8.0f\n", names[e], sum / 256, mn, mx, cyc / 256);
      }
    }
  }
}

// the ViT's QKV GEMM over 1024 frames: heads zero-padded 88 -> 96 (N = 4608, pure 256 x 256 eight-phase tiles) against un-padded (N = 4224 =
// 16 full column tiles + a 128-wide tail on the mixed kernel)
static void qkv_pad_ab(int rounds) {
  const int M = 1024 * 257, K = 1408;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)1 << 24);
  for (auto& v : h) v = (_Float16)d(rng);
  _Float16 *A, *W, *C;
  float* bias;
  const size_t nA = (size_t)M * K, nW = (size_t)4608 * K, nC = (size_t)M * 4608;
  CK(hipMalloc((void**)&A, nA * 2)); CK(hipMalloc((void**)&W, nW * 2)); CK(hipMalloc((void**)&C, nC * 2));
  for (size_t off = 0; off < nA; off += h.size()) CK(hipMemcpy(A + off, h.data(), std::min(h.size(), nA - off) * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), nW * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&bias, 8192 * 4)); CK(hipMemset(bias, 0, 8192 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int N : {4608, 4224}) {
      GemmProb p{};
      p.A = A; p.a = RowView{0, M, K}; p.W = W; p.bias = bias; p.C = C; p.c = RowView{0, M, N}; p.M = M; p.N = N; p.K = K;
      p.tile_cfg = N == 4224 ? 8 : 0;
      double best = 1e30;
      for (int r = 0; r < rounds; ++r) {
        if (launch_gemm(&p, 1, EPI_OP, OP_F16, 0)) { printf("launch failed\n"); return; }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 3; ++i) launch_gemm(&p, 1, EPI_OP, OP_F16, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms / 3);
      }
      printf("qkv N %d: %.3f ms  (%.0f TF/s on executed flops, %.0f on the 4224 useful columns)\n", N, best, 2.0 * M * N * K / best / 1e9, 2.0 * M * 4224 * K / best / 1e9);
    }
}

// Race screen of the eight-phase kernels at full size: the loader-wave kernel accumulates every output in the same order (K tile by K tile,
// two 32-deep MFMA steps each), so the two must agree BIT FOR BIT; a tile read before its DMA landed shows up as a differing element.
static void race_screen(int rounds) {
  const int M = 1024 * 257;
  std::mt19937 rng(9);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)1 << 24);
  for (auto& v : h) v = (_Float16)d(rng);
  _Float16 *A, *W;
  float *bias, *R;
  char *C0, *C1;
  const size_t nA = (size_t)M * 6144, nW = (size_t)9216 * 1408, cbytes = (size_t)M * 9216 * 2;
  CK(hipMalloc((void**)&A, nA * 2)); CK(hipMalloc((void**)&W, nW * 2)); CK(hipMalloc((void**)&C0, cbytes)); CK(hipMalloc((void**)&C1, cbytes));
  for (size_t off = 0; off < nA; off += h.size()) CK(hipMemcpy(A + off, h.data(), std::min(h.size(), nA - off) * 2, hipMemcpyHostToDevice));
  for (size_t off = 0; off < nW; off += h.size()) CK(hipMemcpy(W + off, h.data(), std::min(h.size(), nW - off) * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&bias, 16384 * 4)); CK(hipMemset(bias, 0, 16384 * 4));
  CK(hipMalloc((void**)&R, (size_t)M * 1408 * 4)); CK(hipMemset(R, 0, (size_t)M * 1408 * 4));
  struct S { const char* name; int N, K, epi, cfg; size_t out_bytes; };
  const S shapes[] = {{"kvproj 263168x1408->9216 EPI_KV", 9216, 1408, EPI_KV, 0, (size_t)M * 9216 * 2},
                      {"fc1 263168x1408->6144 EPI_GELU_OP", 6144, 1408, EPI_GELU_OP, 0, (size_t)M * 6144 * 2},
                      {"fc2 263168x6144->1408 EPI_RES_F32, full + tail tiles", 1408, 6144, EPI_RES_F32, 8, (size_t)M * 1408 * 4}};
  std::vector<char> ref, got;
  for (auto& sh : shapes) {
    GemmProb p{};
    p.A = A; p.a = RowView{0, M, sh.K}; p.W = W; p.bias = bias; p.R = R; p.r = RowView{0, M, sh.N};
    p.M = M; p.N = sh.N; p.K = sh.K;
    if (sh.epi == EPI_KV) { p.kv_tokens = M / 32; p.kv_items = 32; p.kv_heads = 12; p.c = RowView{0, 1, 1}; }
    else p.c = RowView{0, M, sh.N};
    // reference: the loader-wave kernel (masked last column tile for N = 1408)
    gemm_set_eight_phase(0);
    GemmProb q = p;
    q.C = C0;
    if (sh.cfg == 8) { q.tile_cfg = 3; q.n_mask = 1; }
    CK(hipMemset(C0, 0, sh.out_bytes));
    if (launch_gemm(&q, 1, sh.epi, OP_F16, 0)) { printf("reference launch failed\n"); return; }
    CK(hipDeviceSynchronize());
    ref.resize(sh.out_bytes);
    CK(hipMemcpy(ref.data(), C0, sh.out_bytes, hipMemcpyDeviceToHost));
    gemm_set_eight_phase(1);
    size_t bad = 0;
    for (int r = 0; r < rounds; ++r) {
      GemmProb e = p;
      e.C = C1;
      if (sh.cfg == 8) e.tile_cfg = 8;
      CK(hipMemset(C1, 0, sh.out_bytes));
      if (launch_gemm(&e, 1, sh.epi, OP_F16, 0)) { printf("eight-phase launch failed\n"); return; }
      CK(hipDeviceSynchronize());
      got.resize(sh.out_bytes);
      CK(hipMemcpy(got.data(), C1, sh.out_bytes, hipMemcpyDeviceToHost));
      if (memcmp(got.data(), ref.data(), sh.out_bytes)) {
        for (size_t i = 0; i < sh.out_bytes; i += 2) bad += got[i] != ref[i] || got[i + 1] != ref[i + 1];
      }
    }
    printf("%-58s %d runs: %s (%zu differing 16-bit words)\n", sh.name, rounds, bad ? "MISMATCH" : "bit-identical to the loader-wave kernel", bad);
    fflush(stdout);
  }
  gemm_set_eight_phase(1);
}

// where the time of the ViT's eight-phase GEMMs goes (256 frames: M = 65792): the same launch with different epilogues and K, so that the
// fixed cost per tile (prologue + epilogue) separates from the K loop: T = rounds x (a + b K)
static void vit_epi_ab(int rounds) {
  const int M = 65792, D = 1408, I = 6144;
  std::mt19937 rng(9);
  std::uniform_real_distribution<float> d(-1.f, 1.f);
  std::vector<_Float16> h((size_t)I * 2 * D);
  for (auto& v : h) v = (_Float16)(0.05f * d(rng));
  _Float16 *A, *W, *C16, *X16;
  float *bias, *X, *stats, *rstat;
  CK(hipMalloc((void**)&A, (size_t)M * I * 2)); CK(hipMemset(A, 0, (size_t)M * I * 2));
  CK(hipMalloc((void**)&W, h.size() * 2)); CK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&C16, (size_t)M * I * 2)); CK(hipMalloc((void**)&X16, (size_t)M * D * 2));
  CK(hipMalloc((void**)&X, (size_t)M * D * 4)); CK(hipMemset(X, 0, (size_t)M * D * 4));
  CK(hipMalloc((void**)&bias, I * 4)); CK(hipMemset(bias, 0, I * 4));
  CK(hipMalloc((void**)&stats, (size_t)M * 11 * 8)); CK(hipMalloc((void**)&rstat, (size_t)M * 8)); CK(hipMemset(rstat, 0, (size_t)M * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, int N, int K, int epi, int tile_cfg) {
    GemmProb p{};
    p.A = A; p.a = RowView{0, M, K}; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.tile_cfg = tile_cfg;
    const bool f32out = epi == EPI_RES_F32 || epi == EPI_F32 || epi == EPI_RES_F32_STAT;
    p.C = f32out ? (void*)X : (void*)C16; p.c = RowView{0, M, N};
    p.R = X; p.r = RowView{0, M, N}; p.aux = C16;
    if (epi == EPI_RES_F32_STAT) { p.ln_y32 = stats; p.ln_y16 = X16; p.ln_y16v = RowView{0, M, N}; }
    if (epi == EPI_LNF_OP || epi == EPI_LNF_GELU_OP) { p.ln_gain = bias; p.ln_y32 = rstat; }
    double best = 1e30;
    for (int r = 0; r < rounds; ++r) {
      if (launch_gemm(&p, 1, epi, OP_F16, 0)) { printf("%-44s refused\n", name); return; }
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) launch_gemm(&p, 1, epi, OP_F16, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, (double)ms / 5 * 1e3);
    }
    const double tiles = tile_cfg == 8 ? 129.0 * 11 : 257.0 * (N / 256);
    printf("%-44s N %4d K %4d  %7.1f us  %6.1f TF/s  %5.1f us per round of 256 tiles\n", name, N, K, best, 2.0 * M * N * K / best * 1e-6, best / (tiles / 256));
  };
  for (int K : {1408, 2816, 6144}) {
    run("N = dim, mixed tiles: EPI_OP (f16 store)", D, K, EPI_OP, 8);
    run("N = dim, mixed tiles: EPI_F32", D, K, EPI_F32, 8);
    run("N = dim, mixed tiles: EPI_RES_F32", D, K, EPI_RES_F32, 8);
    run("N = dim, mixed tiles: EPI_RES_F32_STAT", D, K, EPI_RES_F32_STAT, 8);
    run("N = dim, mixed tiles: EPI_RES_OP", D, K, EPI_RES_OP, 8);
  }
  for (int K : {1408, 2816}) {
    run("fc1: EPI_OP", I, K, EPI_OP, 3);
    run("fc1: EPI_GELU_OP", I, K, EPI_GELU_OP, 3);
    run("fc1: EPI_LNF_GELU_OP", I, K, EPI_LNF_GELU_OP, 3);
    run("qkv: EPI_OP", 4608, K, EPI_OP, 3);
    run("qkv: EPI_LNF_OP", 4608, K, EPI_LNF_OP, 3);
  }
}

// the eight-phase 256 x 256 kernel (8 waves of 128 x 64) against experiment 10 (4 waves of 128 x 128) on the ViT's fc1 / QKV shapes, random
// operands (the clock the chip holds depends on the data), at the 256- and the 1024-frame batch; outputs compared element by element
static void w4_ab(int rounds) {
  std::mt19937 rng(11);
  std::normal_distribution<float> d(0.f, 1.f);
  const size_t Mmax = 263168, Nmax = 6144;
  std::vector<_Float16> ha(Mmax * 1408), hw(Nmax * 2816);
  for (auto& v : ha) v = (_Float16)d(rng);
  for (auto& v : hw) v = (_Float16)(0.05f * d(rng));
  _Float16 *A, *W, *C0, *C1;
  float* bias;
  CK(hipMalloc((void**)&A, ha.size() * 2)); CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&W, hw.size() * 2)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&C0, Mmax * Nmax * 2)); CK(hipMalloc((void**)&C1, Mmax * Nmax * 2));
  CK(hipMalloc((void**)&bias, Nmax * 4)); CK(hipMemset(bias, 0, Nmax * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int K : {1408, 2816})                  // the second K separates the cost per K tile from the fixed cost per output tile
  for (int M : {65792, 263168})
    for (int N : {6144, 4608})
      for (int epi : {EPI_OP, EPI_GELU_OP}) {
        if (epi == EPI_GELU_OP && N != 6144) continue;
        if (K != 1408 && (M != 65792 || N != 6144 || epi != EPI_OP)) continue;
        double t[3] = {0, 0, 0};
        const int variants[3] = {5, 5, 11};          // the middle one: the eight-phase kernel as a persistent workgroup per CU (GemmProb::persist)
        for (int v = 0; v < 3; ++v) {
          gemm_force_variant(variants[v]);
          GemmProb p{};
          p.persist = v == 1;
          p.A = A; p.a = RowView{0, M, K}; p.W = W; p.bias = bias; p.C = v == 1 ? C1 : C0; p.c = RowView{0, M, N}; p.M = M; p.N = N; p.K = K; p.tile_cfg = 3;
          double best = 1e30;
          for (int r = 0; r < rounds; ++r) {
            if (launch_gemm(&p, 1, epi, OP_F16, 0)) { printf("variant %d refused\n", variants[v]); best = -1; break; }
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 3; ++i) launch_gemm(&p, 1, epi, OP_F16, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, (double)ms / 3 * 1e3);
          }
          t[v] = best;
        }
        gemm_force_variant(5);
        // compare a sample of rows (the first 512 and the last 300)
        double worst = 0;
        for (int part = 0; part < 2; ++part) {
          const size_t r0 = part ? (size_t)M - 300 : 0, nr = part ? 300 : 512;
          std::vector<_Float16> c0(nr * N), c1(nr * N);
          CK(hipMemcpy(c0.data(), C0 + r0 * N, nr * N * 2, hipMemcpyDeviceToHost));
          CK(hipMemcpy(c1.data(), C1 + r0 * N, nr * N * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < c0.size(); ++i) worst = std::max(worst, (double)fabsf((float)c0[i] - (float)c1[i]));
        }
        printf("K %4d M %6d N %4d epi %d: eight-phase %8.1f us %7.1f TF/s | persistent %8.1f us %7.1f TF/s | w4p %8.1f us %7.1f TF/s | max |d| (persistent vs the others) %.3e\n", K, M, N, epi, t[0],
               2.0 * M * N * K / t[0] * 1e-6, t[1], 2.0 * M * N * K / t[1] * 1e-6, t[2], 2.0 * M * N * K / t[2] * 1e-6, worst);
      }
}

// stamped timeline of the eight-phase kernel on the ViT's fc1 shape (256 frames): where the ~14 us of fixed cost per output tile go
static void p8_stamp() {
  const int M = 65792, N = 6144, K = 1408;
  std::mt19937 rng(5);
  std::normal_distribution<float> d(0.f, 1.f);
  std::vector<_Float16> ha((size_t)M * K), hw((size_t)N * K);
  for (auto& v : ha) v = (_Float16)d(rng);
  for (auto& v : hw) v = (_Float16)(0.05f * d(rng));
  _Float16 *A, *W, *C;
  float* bias;
  CK(hipMalloc((void**)&A, ha.size() * 2)); CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&W, hw.size() * 2)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&C, (size_t)M * N * 2)); CK(hipMalloc((void**)&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
  const int tiles = 257 * 24;
  unsigned long long* dbg;
  const size_t nent = (size_t)tiles * 8 * 8;
  CK(hipMalloc((void**)&dbg, nent * 8));
  GemmProb p{};
  p.A = A; p.a = RowView{0, M, K}; p.W = W; p.bias = bias; p.C = C; p.c = RowView{0, M, N}; p.M = M; p.N = N; p.K = K; p.tile_cfg = 3;
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 2; ++i) launch_gemm(&p, 1, EPI_OP, OP_F16, 0);
    CK(hipMemset(dbg, 0, nent * 8));
    gemm_set_debug_buffer(dbg);
    launch_gemm(&p, 1, EPI_OP, OP_F16, 0);
    CK(hipDeviceSynchronize());
    gemm_set_debug_buffer(nullptr);
    std::vector<unsigned long long> s(nent);
    CK(hipMemcpy(s.data(), dbg, nent * 8, hipMemcpyDeviceToHost));
    const char* names[7] = {"tile function entered", "addresses + prologue DMA issued", "first K tile landed, barrier passed", "first pair of K tiles done", "last pair begins",
                            "K loop done", "epilogue issued (kernel end)"};
    // intervals of wave 0 and wave 4 of every workgroup, averaged; the time from kernel entry (thread 0) to the tile function as well
    for (int w : {0, 4}) {
      double sum[7] = {0}, entry = 0;
      for (int b = 0; b < tiles; ++b) {
        const unsigned long long* q = &s[((size_t)b * 8 + w) * 8];
        for (int e = 0; e < 7; ++e) sum[e] += (double)(q[e] - q[0]) * 0.01;
        entry += (double)(q[0] - s[((size_t)b * 8) * 8 + 7]) * 0.01;
      }
      printf("wave %d: kernel entry -> tile function %.2f us; then (us after the tile function's start, mean over %d workgroups):\n", w, entry / tiles, tiles);
      for (int e = 0; e < 7; ++e) printf("  %-40s %7.2f\n", names[e], sum[e] / tiles);
    }
    // workgroup turnaround on a CU: end of one workgroup -> start of the next cannot be read from here; the launch's span / rounds is printed instead
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < tiles; ++b) { t0 = std::min(t0, s[((size_t)b * 8) * 8 + 7]); t1 = std::max(t1, s[((size_t)b * 8) * 8 + 6]); }
    printf("launch span %.1f us = %.2f us per round of 256 tiles (%.2f rounds)\n", (double)(t1 - t0) * 0.01, (double)(t1 - t0) * 0.01 / (tiles / 256.0), tiles / 256.0);
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  if (argc > 2 && !strcmp(argv[2], "p8stamp")) { p8_stamp(); return 0; }
  if (argc > 2 && !strcmp(argv[2], "w4")) { w4_ab(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "vitepi")) { vit_epi_ab(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "race")) { race_screen(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "chain")) { chain_ab(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "stamp")) { ring_stamp(); return 0; }
  if (argc > 2 && !strcmp(argv[2], "qkvpad")) { qkv_pad_ab(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "splitk")) { splitk_ab(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "fold")) { fold_ab(rounds); return 0; }
  if (argc > 2 && !strcmp(argv[2], "order")) { order_ab(rounds); return 0; }
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
