// Standalone GPU self-check of the gfx950 kernels against straightforward host loops.
// Test infrastructure: links libmra_hip.so, runs in seconds, no Python.  `kernel_check [quick]`.
// Every case prints one line "ok|FAIL name max_err tol"; exit code = number of failed cases.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "kernels.h"
#include "mra.h"

using namespace mra;

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(99);                                                                            \
    }                                                                                      \
  } while (0)

static int g_fail = 0;
static std::mt19937 g_rng(1234);

static float frand(float s = 1.f) {
  std::normal_distribution<float> d(0.f, s);
  return d(g_rng);
}

// operand-dtype helpers on the host
static uint16_t to_op(float x, int op) {
  if (op == OP_F16) {
    _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
  }
  uint32_t u;
  memcpy(&u, &x, 4);
  u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;  // finite inputs only
  return (uint16_t)u;
}
static float from_op(uint16_t v, int op) {
  if (op == OP_F16) {
    _Float16 h;
    memcpy(&h, &v, 2);
    return (float)h;
  }
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <typename T>
struct Dev {
  T* p = nullptr;
  size_t n = 0;
  explicit Dev(size_t n_) : n(n_) { CK(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T))); }
  Dev(const std::vector<T>& h) : n(h.size()) {
    CK(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)));
    CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  }
  ~Dev() { (void)hipFree(p); }
  std::vector<T> get() const {
    std::vector<T> h(n);
    CK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
  void fill(int byte) { CK(hipMemset(p, byte, n * sizeof(T))); }
};

static void report(const std::string& name, double err, double tol) {
  const bool ok = err <= tol && std::isfinite(err);
  if (!ok) ++g_fail;
  printf("%s %-58s err %.3e tol %.1e\n", ok ? "ok  " : "FAIL", name.c_str(), err, tol);
  fflush(stdout);
}

static long long voff(const RowView& v, int m) { return (long long)(m / v.rpi) * v.item_stride + (long long)(m % v.rpi) * v.ld; }

// ------------------------------------------------------------------------------------------------
static int g_test_order = 0;   // GemmProb::order of the next test_gemm problems (tile walk)
static int g_test_persist = 0; // GemmProb::persist of the next test_gemm problems (one workgroup per CU)
static void test_gemm(int cfg, int epi, int op, int M, int N, int K, bool views, int groups = 1) {
  // activations live in [items, S, K] with the rows of interest at [:, off:off+rpi]
  const int rpi = views ? 5 : (M > 0 ? M : 1);
  const int S = views ? 9 : rpi, off = views ? 3 : 0;
  const int items = (M + rpi - 1) / rpi;
  struct G {
    std::vector<uint16_t> A, W, R16;
    std::vector<float> bias, R;
    RowView av, cv, rv;
    size_t csz;
  };
  std::vector<G> g(groups);
  std::vector<Dev<uint16_t>*> dA, dW, dC16;
  std::vector<Dev<float>*> dB, dR, dC32;
  std::vector<GemmProb> probs(groups);
  const int kvtok = 7, heads = N >= 128 ? 2 : 1;  // EPI_KV: hidden = heads*64, N = sel*hidden
  for (int q = 0; q < groups; ++q) {
    G& x = g[q];
    x.A.resize((size_t)items * S * K);
    for (auto& v : x.A) v = to_op(frand(), op);
    x.W.resize((size_t)N * K);
    for (auto& v : x.W) v = to_op(frand(0.05f), op);
    x.bias.resize(N);
    for (auto& v : x.bias) v = frand(0.5f);
    x.av = RowView{(long long)S * K, rpi, K};
    const int ldc = N + (views ? 8 : 0);
    x.cv = RowView{(long long)S * ldc, rpi, ldc};
    x.rv = RowView{(long long)S * N, rpi, N};
    x.R.resize((size_t)items * S * N);
    for (auto& v : x.R) v = frand();
    x.csz = (size_t)items * S * ldc;
    dA.push_back(new Dev<uint16_t>(x.A));
    dW.push_back(new Dev<uint16_t>(x.W));
    dB.push_back(new Dev<float>(x.bias));
    dR.push_back(new Dev<float>(x.R));
    const size_t kvsz = (size_t)N * ((M + kvtok - 1) / kvtok) * kvtok;
    dC16.push_back(new Dev<uint16_t>(epi == EPI_KV ? kvsz : x.csz));
    dC32.push_back(new Dev<float>(x.csz));
    dC16.back()->fill(0xFF);
    dC32.back()->fill(0xFF);
    if (epi == EPI_RES_OP) {   // in place: C holds the residual in the operand dtype
      x.R16.resize(x.csz);
      for (auto& v : x.R16) v = to_op(frand(), op);
      delete dC16.back();
      dC16.back() = new Dev<uint16_t>(x.R16);
    }
    GemmProb& p = probs[q];
    memset(&p, 0, sizeof(p));
    p.A = dA[q]->p + (size_t)off * K;
    p.a = x.av;
    p.W = dW[q]->p;
    p.bias = dB[q]->p;
    p.M = q == 0 ? M : (M > 3 ? M - 3 : M);
    p.N = N;
    p.K = K;
    p.tile_cfg = cfg >= 0 ? cfg + 1 : 0;   // per problem: the shipped library has no global switches
    p.order = g_test_order; p.persist = g_test_persist;
    if (epi == EPI_RES_F32 || epi == EPI_F32) {
      p.C = dC32[q]->p + (size_t)off * ldc;
      p.c = x.cv;
      p.R = dR[q]->p + (size_t)off * N;
      p.r = x.rv;
    } else if (epi == EPI_KV) {
      p.C = dC16[q]->p;
      p.c = RowView{0, 1, 1};
      p.kv_tokens = kvtok;
      p.kv_items = (M + kvtok - 1) / kvtok;
      p.kv_heads = heads;
    } else {
      p.C = dC16[q]->p + (size_t)off * ldc;
      p.c = x.cv;
      if (epi == EPI_RES_OP) p.aux = p.C;
    }
  }
  const int rc = launch_gemm(probs.data(), groups, epi, op, 0);
  CK(hipDeviceSynchronize());
  double worst = rc == 0 ? 0.0 : 1e30;
  for (int q = 0; q < groups && rc == 0; ++q) {
    const GemmProb& p = probs[q];
    G& x = g[q];
    std::vector<uint16_t> c16 = dC16[q]->get();
    std::vector<float> c32 = dC32[q]->get();
    const int ldc = x.cv.ld;
    for (int m = 0; m < p.M; ++m) {
      const uint16_t* ar = x.A.data() + (size_t)off * K + voff(x.av, m);
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        const uint16_t* wr = x.W.data() + (size_t)n * K;
        for (int k = 0; k < K; ++k) acc += (double)from_op(ar[k], op) * (double)from_op(wr[k], op);
        acc += x.bias[n];
        double got;
        if (epi == EPI_GELU_OP) acc = 0.5 * acc * (1.0 + erf(acc / sqrt(2.0)));
        if (epi == EPI_RES_F32) acc += x.R[(size_t)off * N + voff(x.rv, m) + n];
        if (epi == EPI_RES_OP) acc = from_op(to_op((float)acc, op), op) + from_op(x.R16[(size_t)off * ldc + voff(x.cv, m) + n], op);
        if (epi == EPI_RES_F32 || epi == EPI_F32) {
          got = c32[(size_t)off * ldc + voff(x.cv, m) + n];
        } else if (epi == EPI_KV) {
          const int hidden = heads * 64, sel = n / hidden, within = n % hidden, head = within / 64, d = within % 64;
          const int item = m / kvtok, tok = m % kvtok;
          const size_t dst = ((((size_t)sel * p.kv_items + item) * heads + head) * kvtok + tok) * 64 + d;
          got = from_op(c16[dst], op);
        } else {
          got = from_op(c16[(size_t)off * ldc + voff(x.cv, m) + n], op);
        }
        const double err = fabs(got - acc) / (1.0 + fabs(acc));
        worst = std::isfinite(err) ? std::max(worst, err) : 1e30;
      }
    }
  }
  char name[160];
  snprintf(name, sizeof(name), "gemm cfg%d epi%d %s M%d N%d K%d views%d groups%d", cfg, epi, op == OP_F16 ? "f16" : "bf16", M, N, K,
           (int)views, groups);
  const bool lowp_out = epi == EPI_OP || epi == EPI_GELU_OP || epi == EPI_KV || epi == EPI_RES_OP;
  report(name, worst, op == OP_F16 ? (lowp_out ? 2e-3 : 5e-4) : (lowp_out ? 1.2e-2 : 4e-3));
  for (auto p : dA) delete p;
  for (auto p : dW) delete p;
  for (auto p : dB) delete p;
  for (auto p : dR) delete p;
  for (auto p : dC16) delete p;
  for (auto p : dC32) delete p;
}

// ------------------------------------------------------------------------------------------------
// attention: self (packed QKV rows, ragged mask) or cross (head-major cache)
static void test_attention(int op, int items, int heads, int q_rows, int kv_len, bool self, int nsplit, bool spike = false) {
  const int H = heads * 64;
  std::vector<uint16_t> Q, K, V;
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  std::vector<long long> mask;
  const int ldqkv = 3 * H;
  if (self) {
    Q.resize((size_t)items * q_rows * ldqkv);
    for (auto& v : Q) v = to_op(frand(1.5f), op);
    mask.resize((size_t)items * kv_len, 1);
    for (int n = 0; n < items; ++n) {
      const int valid = 32 + (n * 7) % (kv_len - 32 + 1);  // the 32 queries are never masked
      for (int t = valid; t < kv_len; ++t) mask[(size_t)n * kv_len + t] = 0;
    }
  } else {
    Q.resize((size_t)items * q_rows * H);
    K.resize((size_t)items * heads * kv_len * 64);
    V.resize(K.size());
    for (auto& v : Q) v = to_op(frand(1.5f), op);
    for (auto& v : K) v = to_op(frand(1.5f), op);
    for (auto& v : V) v = to_op(frand(), op);
    if (spike) {  // force a late, large running-max jump (online-softmax rescale path)
      for (int d = 0; d < 64; ++d) {
        const size_t kidx = ((size_t)0 * heads + 0) * kv_len * 64 + (size_t)(kv_len - 5) * 64 + d;
        K[kidx] = to_op(from_op(Q[(size_t)3 * H + d], op) * 3.0f, op);
      }
    }
  }
  Dev<uint16_t> dQ(Q), dK(K), dV(V);
  Dev<long long> dM(mask);
  Dev<uint16_t> dO((size_t)items * q_rows * H);
  dO.fill(0xFF);
  Dev<float> dP(attn_partial_bytes(items, heads, q_rows, nsplit) / 4 + 4);
  a.Q = dQ.p;
  a.O = dO.p;
  a.o_item_stride = (long long)q_rows * H;
  a.o_ld = H;
  if (self) {
    a.K = dQ.p + H;
    a.V = dQ.p + 2 * H;
    a.q_item_stride = a.k_item_stride = a.v_item_stride = (long long)q_rows * ldqkv;
    a.q_ld = a.k_ld = a.v_ld = ldqkv;
    a.k_head_stride = a.v_head_stride = 64;
    a.mask = dM.p;
    a.mask_ld = kv_len;
  } else {
    a.K = dK.p;
    a.V = dV.p;
    a.q_item_stride = (long long)q_rows * H;
    a.q_ld = H;
    a.k_item_stride = a.v_item_stride = (long long)heads * kv_len * 64;
    a.k_head_stride = a.v_head_stride = (long long)kv_len * 64;
    a.k_ld = a.v_ld = 64;
  }
  a.items = items;
  a.heads = heads;
  a.q_rows = q_rows;
  a.kv_len = kv_len;
  a.scale = 0.125f;
  a.nsplit = nsplit;
  a.part = nsplit > 1 ? dP.p : nullptr;
  const int rc = launch_attention(a, op, 0);
  CK(hipDeviceSynchronize());
  double worst = rc == 0 ? 0.0 : 1e30;
  if (rc == 0) {
    std::vector<uint16_t> O = dO.get();
    std::vector<double> s(kv_len);
    for (int n = 0; n < items; ++n)
      for (int h = 0; h < heads; ++h)
        for (int q = 0; q < q_rows; ++q) {
          const uint16_t* qp = self ? Q.data() + ((size_t)n * q_rows + q) * ldqkv + h * 64 : Q.data() + ((size_t)n * q_rows + q) * H + h * 64;
          double mx = -1e300;
          for (int t = 0; t < kv_len; ++t) {
            const uint16_t* kp = self ? Q.data() + ((size_t)n * q_rows + t) * ldqkv + H + h * 64
                                      : K.data() + (((size_t)n * heads + h) * kv_len + t) * 64;
            double acc = 0;
            for (int d = 0; d < 64; ++d) acc += (double)from_op(qp[d], op) * from_op(kp[d], op);
            acc *= 0.125;
            if (self) acc += (1.0 - (double)mask[(size_t)n * kv_len + t]) * -10000.0;
            s[t] = acc;
            mx = std::max(mx, acc);
          }
          double den = 0;
          for (int t = 0; t < kv_len; ++t) { s[t] = exp(s[t] - mx); den += s[t]; }
          for (int d = 0; d < 64; ++d) {
            double acc = 0;
            for (int t = 0; t < kv_len; ++t) {
              const uint16_t* vp = self ? Q.data() + ((size_t)n * q_rows + t) * ldqkv + 2 * H + h * 64
                                        : V.data() + (((size_t)n * heads + h) * kv_len + t) * 64;
              acc += s[t] * from_op(vp[d], op);
            }
            acc /= den;
            const double got = from_op(O[((size_t)n * q_rows + q) * H + h * 64 + d], op);
            const double err = fabs(got - acc);
            worst = std::isfinite(err) ? std::max(worst, err) : 1e30;
          }
        }
  }
  char name[160];
  snprintf(name, sizeof(name), "attention %s %s items%d heads%d q%d kv%d split%d%s", self ? "self" : "cross", op == OP_F16 ? "f16" : "bf16",
           items, heads, q_rows, kv_len, nsplit, spike ? " spike" : "");
  report(name, worst, op == OP_F16 ? 3e-3 : 2e-2);
}

// P . enc form: W given K-major ([k_rows][ldw] row-major, columns = output n), K padded beyond k_rows where A is zero
static void test_gemm_kmajor(int op, int M, int N, int K, int k_rows, int batch) {
  const int ldw = N + 16;
  std::vector<uint16_t> A((size_t)batch * M * K), W((size_t)batch * k_rows * ldw);
  for (size_t b = 0; b < (size_t)batch; ++b)
    for (int m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k) A[(b * M + m) * K + k] = to_op(k < k_rows ? frand() : 0.f, op);
  for (auto& v : W) v = to_op(frand(0.05f), op);
  Dev<uint16_t> dA(A), dW(W), dC((size_t)batch * M * N);
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.A = dA.p; p.a = RowView{0, M, K}; p.W = dW.p; p.C = dC.p; p.c = RowView{0, M, N};
  p.M = M; p.N = N; p.K = K; p.batch = batch; p.a_bs = (long long)M * K; p.w_bs = (long long)k_rows * ldw; p.c_bs_bytes = (long long)M * N * 2;
  p.w_ld = ldw; p.k_rows = k_rows; p.tile_cfg = 5;
  const int rc = launch_gemm(&p, 1, EPI_OP, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> c = dC.get();
  double worst = rc ? 1e30 : 0;
  for (int b = 0; b < batch && !rc; ++b)
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < k_rows; ++k) acc += (double)from_op(A[((size_t)b * M + m) * K + k], op) * from_op(W[((size_t)b * k_rows + k) * ldw + n], op);
        worst = std::max(worst, fabs(from_op(c[((size_t)b * M + m) * N + n], op) - acc) / (1 + fabs(acc)));
      }
  char name[128];
  snprintf(name, sizeof(name), "gemm K-major W %s M%d N%d K%d (rows %d) x%d", op == OP_F16 ? "f16" : "bf16", M, N, K, k_rows, batch);
  report(name, worst, 3e-3);
}

// EPI_RES_LN on the 96 x 64 ring tile: C = A W^T + bias + R, then LayerNorm of every finished row by the last-arriving column tile of its 64-row
// block.  Launched twice on the same counters (they must be back at zero), two problems with their own gains, ragged M.
static void test_gemm_res_ln(int op, int M, int N, int K, int groups) {
  struct G { std::vector<uint16_t> A, W; std::vector<float> bias, R, gain, beta; };
  std::vector<G> g(groups);
  std::vector<Dev<uint16_t>*> dA, dW, dY16;
  std::vector<Dev<float>*> dB, dR, dC, dY32, dG, dBe;
  const int mt = (M + 63) / 64;
  Dev<unsigned> dCnt((size_t)groups * mt);
  CK(hipMemset(dCnt.p, 0, (size_t)groups * mt * 4));
  std::vector<GemmProb> probs(groups);
  for (int q = 0; q < groups; ++q) {
    G& x = g[q];
    const int Mq = q == 0 ? M : M - 7;
    x.A.resize((size_t)M * K); x.W.resize((size_t)N * K); x.bias.resize(N); x.R.resize((size_t)M * N); x.gain.resize(N); x.beta.resize(N);
    for (auto& v : x.A) v = to_op(frand(), op);
    for (auto& v : x.W) v = to_op(frand(0.05f), op);
    for (auto& v : x.bias) v = frand(0.5f);
    for (auto& v : x.R) v = frand(2.f);
    for (auto& v : x.gain) v = 1.f + frand(0.3f);
    for (auto& v : x.beta) v = frand(0.2f);
    dA.push_back(new Dev<uint16_t>(x.A)); dW.push_back(new Dev<uint16_t>(x.W)); dB.push_back(new Dev<float>(x.bias)); dR.push_back(new Dev<float>(x.R));
    dG.push_back(new Dev<float>(x.gain)); dBe.push_back(new Dev<float>(x.beta));
    dC.push_back(new Dev<float>((size_t)M * N)); dY32.push_back(new Dev<float>((size_t)M * N)); dY16.push_back(new Dev<uint16_t>((size_t)M * N));
    dY32.back()->fill(0xFF); dY16.back()->fill(0xFF);
    GemmProb& p = probs[q];
    memset(&p, 0, sizeof(p));
    p.A = dA[q]->p; p.a = RowView{0, M, K}; p.W = dW[q]->p; p.bias = dB[q]->p; p.R = dR[q]->p; p.r = RowView{0, M, N};
    p.C = dC[q]->p; p.c = RowView{0, M, N}; p.M = Mq; p.N = N; p.K = K; p.tile_cfg = 11;
    p.ln_gain = dG[q]->p; p.ln_bias = dBe[q]->p; p.ln_eps = 1e-12f; p.ln_y32 = dY32[q]->p; p.ln_y32v = RowView{0, M, N};
    p.ln_y16 = dY16[q]->p; p.ln_y16v = RowView{0, M, N}; p.ln_counter = dCnt.p + (size_t)q * mt;
  }
  int rc = launch_gemm(probs.data(), groups, EPI_RES_LN, op, 0);
  if (!rc) rc = launch_gemm(probs.data(), groups, EPI_RES_LN, op, 0);   // the counters were left at zero
  CK(hipDeviceSynchronize());
  double worst = rc ? 1e30 : 0;
  std::vector<unsigned> cnt = dCnt.get();
  for (auto v : cnt) if (v != 0) worst = 1e30;
  for (int q = 0; q < groups && !rc; ++q) {
    const G& x = g[q];
    const int Mq = probs[q].M;
    std::vector<float> y32 = dY32[q]->get();
    std::vector<uint16_t> y16 = dY16[q]->get();
    std::vector<double> row(N);
    for (int m = 0; m < Mq; ++m) {
      double mean = 0;
      for (int n = 0; n < N; ++n) {
        double acc = x.bias[n] + x.R[(size_t)m * N + n];
        for (int k = 0; k < K; ++k) acc += (double)from_op(x.A[(size_t)m * K + k], op) * from_op(x.W[(size_t)n * K + k], op);
        row[n] = acc; mean += acc;
      }
      mean /= N;
      double var = 0;
      for (int n = 0; n < N; ++n) var += (row[n] - mean) * (row[n] - mean);
      const double rstd = 1.0 / sqrt(var / N + 1e-12);
      for (int n = 0; n < N; ++n) {
        const double want = (row[n] - mean) * rstd * x.gain[n] + x.beta[n];
        worst = std::max(worst, fabs(y32[(size_t)m * N + n] - want) / (1 + fabs(want)));
        worst = std::max(worst, 0.1 * fabs(from_op(y16[(size_t)m * N + n], op) - want) / (1 + fabs(want)));   // the 16-bit copy at a 10 x bar
      }
    }
  }
  char name[128];
  snprintf(name, sizeof(name), "gemm residual + LayerNorm in one launch %s M%d N%d K%d groups%d", op == OP_F16 ? "f16" : "bf16", M, N, K, groups);
  report(name, worst, op == OP_F16 ? 5e-4 : 4e-3);
  for (auto p : dA) delete p; for (auto p : dW) delete p; for (auto p : dY16) delete p;
  for (auto p : dB) delete p; for (auto p : dR) delete p; for (auto p : dC) delete p; for (auto p : dY32) delete p; for (auto p : dG) delete p; for (auto p : dBe) delete p;
}

// LayerNorm folded into the GEMMs around it (EPI_RES_F32_STAT -> group statistics -> EPI_LNF_*): producer x = A1 W1^T + b1 + R with the
// op-dtype copy and the 128-column group statistics; then y = LN(x; gain, beta) W2^T + b2 computed as the consumer GEMM over the raw copy
// with W2 diag(gain), its row sums and b2 + W2 beta (prepared here on the host exactly as vit_fold_weight_kernel does).  Reference in double:
// the explicit LayerNorm of the fp32 rows, then the product with the UNfolded W2.
static void test_gemm_ln_fold(int op, int M, int D, int K1, int N2, bool gelu, float offset) {
  std::vector<uint16_t> A1((size_t)M * K1), W1((size_t)D * K1), W2((size_t)N2 * D), W2f((size_t)N2 * D);
  for (auto& v : A1) v = to_op(frand(), op);
  for (auto& v : W1) v = to_op(frand(0.05f), op);
  for (auto& v : W2) v = to_op(frand(0.05f), op);
  std::vector<float> b1(D), R((size_t)M * D), gain(D), beta(D), b2(N2), cs(N2), bf(N2);
  for (auto& v : b1) v = frand(0.5f);
  for (auto& v : R) v = offset + frand(2.f);      // offset: a row mean far from zero (the case a sum / sum-of-squares formulation would lose)
  for (auto& v : gain) v = 1.f + frand(0.3f);
  for (auto& v : beta) v = frand(0.2f);
  for (auto& v : b2) v = frand(0.5f);
  for (int n = 0; n < N2; ++n) {
    float c = 0.f, b = 0.f;
    for (int k = 0; k < D; ++k) {
      const float w = from_op(W2[(size_t)n * D + k], op);
      W2f[(size_t)n * D + k] = to_op(w * gain[k], op);
      c += from_op(W2f[(size_t)n * D + k], op);
      b += beta[k] * w;
    }
    cs[n] = c; bf[n] = b + b2[n];
  }
  const int G = D / 128;
  Dev<uint16_t> dA1(A1), dW1(W1), dW2f(W2f), dX16((size_t)M * D), dY((size_t)M * N2);
  Dev<float> dB1(b1), dR(R), dX((size_t)M * D), dGroups((size_t)M * G * 2), dStat((size_t)M * 2), dCs(cs), dBf(bf);
  dX16.fill(0xFF); dGroups.fill(0xFF);
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.A = dA1.p; p.a = RowView{0, M, K1}; p.W = dW1.p; p.bias = dB1.p; p.R = dR.p; p.r = RowView{0, M, D};
  p.C = dX.p; p.c = RowView{0, M, D}; p.M = M; p.N = D; p.K = K1; p.tile_cfg = D % 256 ? 8 : 3;
  p.ln_y32 = dGroups.p; p.ln_y16 = dX16.p; p.ln_y16v = RowView{0, M, D};
  int rc = launch_gemm(&p, 1, EPI_RES_F32_STAT, op, 0);
  CK(hipDeviceSynchronize());
  // the group statistics -> (mean, rstd) per row: host restatement of vit_group_stats_kernel (the kernel itself is exercised by the ViT tests)
  std::vector<float> groups = dGroups.get(), x = dX.get();
  std::vector<uint16_t> x16 = dX16.get();
  std::vector<float> stat((size_t)M * 2);
  double worst = rc ? 1e30 : 0, worst_stat = 0;
  const float eps = 1e-6f;
  for (int m = 0; m < M && !rc; ++m) {
    float mean = 0.f, m2 = 0.f;
    for (int g = 0; g < G; ++g) mean += groups[((size_t)m * G + g) * 2];
    mean /= G;
    for (int g = 0; g < G; ++g) { const float d = groups[((size_t)m * G + g) * 2] - mean; m2 += groups[((size_t)m * G + g) * 2 + 1] + 128.f * d * d; }
    stat[(size_t)m * 2] = mean; stat[(size_t)m * 2 + 1] = 1.0f / sqrtf(m2 / D + eps);
    double dm = 0, dv = 0;
    for (int n = 0; n < D; ++n) dm += x[(size_t)m * D + n];
    dm /= D;
    for (int n = 0; n < D; ++n) dv += (x[(size_t)m * D + n] - dm) * (x[(size_t)m * D + n] - dm);
    worst_stat = std::max(worst_stat, fabs(mean - dm) / (1 + fabs(dm)));
    worst_stat = std::max(worst_stat, fabs(stat[(size_t)m * 2 + 1] - 1.0 / sqrt(dv / D + eps)) * sqrt(dv / D + eps));
    for (int n = 0; n < D; ++n) {
      double acc = b1[n] + R[(size_t)m * D + n];
      for (int k = 0; k < K1; ++k) acc += (double)from_op(A1[(size_t)m * K1 + k], op) * from_op(W1[(size_t)n * K1 + k], op);
      worst = std::max(worst, fabs(x[(size_t)m * D + n] - acc) / (1 + fabs(acc)));
      if (x16[(size_t)m * D + n] != to_op(x[(size_t)m * D + n], op)) worst = 1e30;   // the copy is the stored fp32 value rounded once
    }
  }
  char name[160];
  snprintf(name, sizeof(name), "gemm residual + op-dtype copy + group statistics %s M%d N%d K%d offset %.0f", op == OP_F16 ? "f16" : "bf16", M, D, K1, offset);
  report(name, std::max(worst, worst_stat * 10), 2e-4);
  if (rc) return;
  Dev<float> dSt(stat);
  GemmProb q;
  memset(&q, 0, sizeof(q));
  q.A = dX16.p; q.a = RowView{0, M, D}; q.W = dW2f.p; q.bias = dBf.p; q.C = dY.p; q.c = RowView{0, M, N2}; q.M = M; q.N = N2; q.K = D; q.tile_cfg = 3;
  q.ln_gain = dCs.p; q.ln_y32 = dSt.p;
  rc = launch_gemm(&q, 1, gelu ? EPI_LNF_GELU_OP : EPI_LNF_OP, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> y = dY.get();
  worst = rc ? 1e30 : 0;
  std::vector<double> ln(D);
  for (int m = 0; m < M && !rc; ++m) {
    double dm = 0, dv = 0;
    for (int n = 0; n < D; ++n) dm += x[(size_t)m * D + n];
    dm /= D;
    for (int n = 0; n < D; ++n) dv += (x[(size_t)m * D + n] - dm) * (x[(size_t)m * D + n] - dm);
    const double rstd = 1.0 / sqrt(dv / D + eps);
    for (int n = 0; n < D; ++n) ln[n] = (x[(size_t)m * D + n] - dm) * rstd * gain[n] + beta[n];
    for (int n = 0; n < N2; ++n) {
      double acc = b2[n];
      for (int k = 0; k < D; ++k) acc += ln[k] * from_op(W2[(size_t)n * D + k], op);
      if (gelu) acc = 0.5 * acc * (1.0 + erf(acc * 0.70710678118654752440));
      worst = std::max(worst, fabs(from_op(y[(size_t)m * N2 + n], op) - acc) / (1 + fabs(acc)));
    }
  }
  snprintf(name, sizeof(name), "gemm with the LayerNorm folded in (%s) %s M%d N%d K%d offset %.0f", gelu ? "GELU" : "plain", op == OP_F16 ? "f16" : "bf16", M, N2, D, offset);
  // operands: raw rows and W diag(gain) rounded to the operand dtype (the separate-LayerNorm form rounds LN(x) and W instead): same order of error
  report(name, worst, op == OP_F16 ? 4e-3 * (1 + offset / 4) : 3e-2 * (1 + offset / 4));
}

// EPI_RES_OP_STAT: the operand-dtype residual stream updated in place exactly as EPI_RES_OP does (bit for bit), plus per (row, 64 columns)
// the mean and the squared deviations of the values as stored
static void test_gemm_res_op_stat(int op, int M, int N, int K, float offset) {
  std::vector<uint16_t> A((size_t)M * K), W((size_t)N * K), X((size_t)M * N);
  for (auto& v : A) v = to_op(frand(), op);
  for (auto& v : W) v = to_op(frand(0.05f), op);
  for (auto& v : X) v = to_op(offset + frand(2.f), op);
  std::vector<float> bias(N);
  for (auto& v : bias) v = frand(0.5f);
  const int G = N / 64;
  Dev<uint16_t> dA(A), dW(W), dX0(X), dX1(X);
  Dev<float> dB(bias), dG((size_t)M * G * 2);
  dG.fill(0xFF);
  int rc = 0;
  for (int pass = 0; pass < 2 && !rc; ++pass) {
    GemmProb p;
    memset(&p, 0, sizeof(p));
    uint16_t* x = pass ? dX1.p : dX0.p;
    p.A = dA.p; p.a = RowView{0, M, K}; p.W = dW.p; p.bias = dB.p; p.C = x; p.aux = x; p.c = RowView{0, M, N};
    p.M = M; p.N = N; p.K = K; p.tile_cfg = N % 256 ? 8 : 3; p.ln_y32 = dG.p;
    rc = launch_gemm(&p, 1, pass ? EPI_RES_OP_STAT : EPI_RES_OP, op, 0);
  }
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> x0 = dX0.get(), x1 = dX1.get();
  std::vector<float> g = dG.get();
  double worst = rc ? 1e30 : 0;
  for (size_t i = 0; i < x0.size() && !rc; ++i) if (x0[i] != x1[i]) { worst = 1e30; break; }
  for (int m = 0; m < M && !rc; ++m)
    for (int b = 0; b < G; ++b) {
      double mean = 0, q = 0;
      for (int n = 0; n < 64; ++n) mean += from_op(x1[(size_t)m * N + b * 64 + n], op);
      mean /= 64;
      for (int n = 0; n < 64; ++n) { const double d = from_op(x1[(size_t)m * N + b * 64 + n], op) - mean; q += d * d; }
      worst = std::max(worst, fabs(g[((size_t)m * G + b) * 2] - mean) / (1 + fabs(mean)));
      worst = std::max(worst, fabs(g[((size_t)m * G + b) * 2 + 1] - q) / (1 + q));
    }
  char name[160];
  snprintf(name, sizeof(name), "gemm operand-dtype residual + 64-column statistics %s M%d N%d K%d offset %.0f", op == OP_F16 ? "f16" : "bf16", M, N, K, offset);
  report(name, worst, 1e-5);
}

// n_mask: N not a multiple of the tile, plain [M][N] output rows; columns past N are neither read (bias, residual) nor stored
static void test_gemm_masked(int cfg, int epi, int op, int M, int N, int K) {
  std::vector<uint16_t> A((size_t)M * K), W((size_t)N * K);
  for (auto& v : A) v = to_op(frand(), op);
  for (auto& v : W) v = to_op(frand(0.05f), op);
  std::vector<float> bias(N), R((size_t)M * N), C0((size_t)M * N + 512, 7.5f);
  for (auto& v : bias) v = frand(0.5f);
  for (auto& v : R) v = frand();
  Dev<uint16_t> dA(A), dW(W);
  Dev<float> dB(bias), dR(R), dC(C0);
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.A = dA.p; p.a = RowView{0, M, K}; p.W = dW.p; p.bias = dB.p; p.R = dR.p; p.r = RowView{0, M, N};
  p.C = dC.p; p.c = RowView{0, M, N}; p.M = M; p.N = N; p.K = K; p.n_mask = 1; p.tile_cfg = cfg + 1; p.order = g_test_order; p.persist = g_test_persist;
  const int rc = launch_gemm(&p, 1, epi, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> c = dC.get();
  double worst = rc ? 1e30 : 0;
  for (int m = 0; m < M && !rc; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = bias[n] + (epi == EPI_RES_F32 ? R[(size_t)m * N + n] : 0.0);
      for (int k = 0; k < K; ++k) acc += (double)from_op(A[(size_t)m * K + k], op) * from_op(W[(size_t)n * K + k], op);
      worst = std::max(worst, fabs(c[(size_t)m * N + n] - acc) / (1 + fabs(acc)));
    }
  for (size_t i = (size_t)M * N; i < c.size(); ++i) if (c[i] != 7.5f) worst = 1e30;   // nothing may be written past the matrix
  char name[128];
  snprintf(name, sizeof(name), "gemm masked-N cfg%d epi%d %s M%d N%d K%d", cfg, epi, op == OP_F16 ? "f16" : "bf16", M, N, K);
  report(name, worst, 2e-4);
}

// P . enc with the row factors of the split softmax applied inside the GEMM (GemmProb::pscale): against the rescale-pass arithmetic --
// P[m][k] = op(float(P~[m][k]) * g[tile(k)][m]), then the product in double.  A is zero from column kv on; the factor slots past the last tile
// reuse its (finite) factors.
static void test_gemm_pscale(int op, int M, int N, int kv, int batch) {
  const int ntiles = (kv + 175) / 176, K = ((std::max((kv + 127) / 128 * 128, ntiles * 176) + 127) / 128) * 128, ldw = N;
  std::vector<uint16_t> A((size_t)batch * M * K), W((size_t)batch * kv * ldw);
  std::vector<float> G((size_t)batch * ntiles * 512, 123.f);
  for (size_t b = 0; b < (size_t)batch; ++b)
    for (int m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k) A[(b * M + m) * K + k] = k < kv ? to_op(fabsf(frand()), op) : to_op(0.f, op);   // zero from column kv on (the caller's contract)
  for (auto& v : W) v = to_op(frand(0.5f), op);
  for (size_t b = 0; b < (size_t)batch; ++b)
    for (int t = 0; t < ntiles; ++t)
      for (int m = 0; m < M; ++m) G[(b * ntiles + t) * 512 + m] = expf(-4.f * fabsf(frand())) * (t % 3 == 2 ? 1e-6f : 1.f);
  Dev<uint16_t> dA(A), dW(W), dC((size_t)batch * M * N);
  Dev<float> dG(G);
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.A = dA.p; p.a = RowView{0, M, K}; p.W = dW.p; p.C = dC.p; p.c = RowView{0, M, N};
  p.M = M; p.N = N; p.K = K; p.batch = batch; p.a_bs = (long long)M * K; p.w_bs = (long long)kv * ldw; p.c_bs_bytes = (long long)M * N * 2;
  p.w_ld = ldw; p.k_rows = kv; p.tile_cfg = 5; p.pscale = dG.p; p.ps_ntiles = ntiles;
  const int rc = launch_gemm(&p, 1, EPI_OP, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> c = dC.get();
  double worst = rc ? 1e30 : 0;
  for (int b = 0; b < batch && !rc; ++b)
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; n += 7) {
        double acc = 0;
        for (int k = 0; k < kv; ++k) {
          const float gk = G[((size_t)b * ntiles + k / 176) * 512 + m];   // f16: the kernel rounds the factor to f16 first (packed multiplies)
          const float pr = from_op(to_op(from_op(A[((size_t)b * M + m) * K + k], op) * (op == OP_F16 ? from_op(to_op(gk, op), op) : gk), op), op);
          acc += (double)pr * from_op(W[((size_t)b * kv + k) * ldw + n], op);
        }
        const double got = from_op(c[((size_t)b * M + m) * N + n], op);
        const double err = fabs(got - acc) / (1 + fabs(acc));
        worst = std::isfinite(err) ? std::max(worst, err) : 1e30;
      }
  char name[128];
  snprintf(name, sizeof(name), "gemm K-major W + row factors %s M%d N%d kv%d (K %d) x%d", op == OP_F16 ? "f16" : "bf16", M, N, kv, K, batch);
  report(name, worst, op == OP_F16 ? 2e-3 : 1.2e-2);
}

// scores of the folded path: batched, ragged N, EPI_SOFTPART -- P~[m][n] = exp2(alpha s - tile max) in the operand dtype, tile maxima and
// tile sums (of the ROUNDED P~); columns past N inside the last tile are zero.
static void test_gemm_softpart(int op, int M, int kv, int K, int batch) {
  const int ntiles = (kv + 175) / 176, ldp = ntiles * 176 + 16;
  const float alpha = 0.125f * 1.4426950408889634f;
  std::vector<uint16_t> A((size_t)batch * M * K), W((size_t)batch * kv * K);
  for (auto& v : A) v = to_op(frand(), op);
  for (auto& v : W) v = to_op(frand(), op);
  Dev<uint16_t> dA(A), dW(W), dP((size_t)batch * M * ldp);
  Dev<float> dM((size_t)batch * M * ntiles), dL((size_t)batch * M * ntiles);
  dP.fill(0xFF);
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.A = dA.p; p.a = RowView{0, M, K}; p.a_bs = (long long)M * K; p.W = dW.p; p.w_bs = (long long)kv * K;
  p.M = M; p.N = kv; p.K = K; p.batch = batch; p.n_ragged = 1; p.tile_cfg = 5;
  p.C = dP.p; p.c = RowView{0, M, ldp}; p.c_bs_bytes = (long long)M * ldp * 2; p.alpha = alpha; p.stat_m = dM.p; p.stat_l = dL.p;
  const int rc = launch_gemm(&p, 1, EPI_SOFTPART, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> P = dP.get();
  std::vector<float> sm = dM.get(), sl = dL.get();
  double worst = rc ? 1e30 : 0;
  std::vector<double> srow(ntiles * 176);
  for (int b = 0; b < batch && !rc; ++b)
    for (int m = 0; m < M; ++m) {
      for (int n = 0; n < kv; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)from_op(A[((size_t)b * M + m) * K + k], op) * from_op(W[((size_t)b * kv + n) * K + k], op);
        srow[n] = acc * alpha;
      }
      for (int t = 0; t < ntiles; ++t) {
        const int lo = t * 176, hi = std::min(kv, lo + 176);
        double mx = -1e300, l = 0;
        for (int n = lo; n < hi; ++n) mx = std::max(mx, srow[n]);
        const size_t si = ((size_t)b * M + m) * ntiles + t;
        worst = std::max(worst, fabs(sm[si] - mx) / (1 + fabs(mx)));
        for (int n = lo; n < lo + 176; ++n) {
          const double got = from_op(P[((size_t)b * M + m) * ldp + n], op);
          const double want = n < hi ? exp2(srow[n] - sm[si]) : 0.0;     // relative to the kernel's own maximum
          l += got;
          const double err = fabs(got - want);
          worst = std::isfinite(err) ? std::max(worst, err) : 1e30;
        }
        worst = std::max(worst, fabs(sl[si] - l) / (1 + fabs(l)) * 10);   // the sum is of the rounded values: tight
      }
    }
  char name[128];
  snprintf(name, sizeof(name), "gemm softmax-partial %s M%d kv%d K%d x%d", op == OP_F16 ? "f16" : "bf16", M, kv, K, batch);
  report(name, worst, op == OP_F16 ? 2e-3 : 1.2e-2);
}

// batched launch (one weight matrix per batch entry) with a ragged N: the folded cross-attention's GEMMs
// wrap: GemmProb::w_kwrap -- W holds only K / 2 columns and is walked twice (A = (hi | lo) rows of the split-precision scores product);
// W is allocated EXACTLY (a stride of K instead of K / 2 would run off its end)
static void test_gemm_batched(int cfg, int epi, int op, int M, int N, int K, int batch, bool ragged, bool wrap = false) {
  const int t = cfg == 0 ? 64 : (cfg == 2 ? 256 : (cfg == 4 ? 176 : 128));   // weight rows per tile (config 3: 128 x 384, 4: 176 x 384)
  const int ldc = (N + t - 1) / t * t;                       // C rows hold whole tiles
  const int KW = wrap ? K / 2 : K;
  std::vector<uint16_t> A((size_t)batch * M * K), W((size_t)batch * N * KW);
  for (auto& v : A) v = to_op(frand(), op);
  for (auto& v : W) v = to_op(frand(0.05f), op);
  std::vector<float> bias((size_t)batch * N);
  for (auto& v : bias) v = ragged ? 0.f : frand(0.5f);
  Dev<uint16_t> dA(A), dW(W), dC16((size_t)batch * M * ldc);
  Dev<float> dB(bias), dC32((size_t)batch * M * ldc);
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.A = dA.p; p.a = RowView{0, M, K}; p.W = dW.p; p.bias = ragged ? nullptr : dB.p;
  p.M = M; p.N = N; p.K = K;
  p.batch = batch; p.a_bs = (long long)M * K; p.w_bs = (long long)N * KW; p.bias_bs = N;
  p.n_ragged = ragged ? 1 : 0; p.tile_cfg = cfg + 1; p.w_kwrap = wrap ? KW / 64 : 0;
  const bool f32 = epi == EPI_F32;
  p.C = f32 ? (void*)dC32.p : (void*)dC16.p; p.c = RowView{0, M, ldc};
  p.c_bs_bytes = (long long)M * ldc * (f32 ? 4 : 2);
  const int rc = launch_gemm(&p, 1, epi, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> c16 = dC16.get();
  std::vector<float> c32 = dC32.get();
  double worst = rc ? 1e30 : 0;
  for (int b = 0; b < batch && !rc; ++b)
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double acc = bias[(size_t)b * N + n];
        for (int k = 0; k < K; ++k) acc += (double)from_op(A[((size_t)b * M + m) * K + k], op) * from_op(W[((size_t)b * N + n) * KW + k % KW], op);
        const size_t ci = ((size_t)b * M + m) * ldc + n;
        const double got = f32 ? c32[ci] : from_op(c16[ci], op);
        worst = std::max(worst, fabs(got - acc) / (1 + fabs(acc)));
      }
  char name[128];
  snprintf(name, sizeof(name), "gemm batched cfg%d epi%d M%d N%d K%d x%d%s%s", cfg, epi, M, N, K, batch, ragged ? " ragged-N" : "", wrap ? " W walked twice" : "");
  report(name, worst, f32 ? 2e-4 : 3e-3);
}

// Folded cross-attention on the streaming kernels (fold_stream.hip): scores with split-softmax statistics, row statistics,
// P . enc with the tile factors applied in registers -- against softmax(alpha' Q' enc^T) enc in double on the same f16 inputs.
// `gain` scales Q' so that rows are flat (1) or peaked (tile maxima far apart: factors down to 2^-24 and exact zeros).
static void test_fold_stream(int items, int kv, int E, float gain) {
  const int R = 384, op = OP_F16;
  const int kvp = (std::max((kv + 127) / 128 * 128, (kv + 175) / 176 * 176) + 127) / 128 * 128;
  if (!fold_stream_supported(R, E, kv, kvp, op)) { report("fold stream: shape not supported", 1e30, 0); return; }
  const int sld = fold_stream_stat_ld(kvp);
  std::vector<uint16_t> Q((size_t)items * R * E), X((size_t)items * kv * E);
  for (auto& v : Q) v = to_op(frand(0.3f) * gain, op);
  for (auto& v : X) v = to_op(frand(), op);
  Dev<uint16_t> dQ(Q), dQb((size_t)items * R * E), dX(X), dP((size_t)items * R * kvp), dU((size_t)items * R * E), dG((size_t)items * R * sld);
  Dev<float> dM((size_t)items * R * sld), dL((size_t)items * R * sld), dI((size_t)items * R);
  CK(hipMemset(dP.p, 0xff, (size_t)items * R * kvp * 2));     // NaN patterns: every P~ column the pv product reads must have been written
  FoldStreamArgs a;
  memset(&a, 0, sizeof(a));
  a.qp = dQ.p; a.qpb = dQb.p; a.enc = dX.p; a.p = dP.p; a.u = dU.p; a.stat_m = dM.p; a.stat_l = dL.p; a.gexp = dG.p; a.ginv = dI.p;
  a.items = items; a.kv = kv; a.kvp = kvp; a.E = E; a.alpha = 0.125f * 1.4426950408889634f; a.phase = 3;
  const int rc = launch_fold_stream(a, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> u = dU.get();
  double worst = rc ? 1e30 : 0, peak = 0;
  std::vector<double> s(kv), ref(E);
  for (int it = 0; it < items && !rc; ++it)
    for (int r = 0; r < R; r += (items * R > 800 ? 7 : 1)) {     // every 7th row on the larger cases
      double mx = -1e300;
      for (int k = 0; k < kv; ++k) {
        double acc = 0;
        for (int e = 0; e < E; ++e) acc += (double)from_op(Q[((size_t)it * R + r) * E + e], op) * from_op(X[((size_t)it * kv + k) * E + e], op);
        s[k] = acc * 0.125;
        mx = std::max(mx, s[k]);
      }
      double L = 0, pm = 0;
      for (int k = 0; k < kv; ++k) { s[k] = exp(s[k] - mx); L += s[k]; }
      std::fill(ref.begin(), ref.end(), 0.0);
      for (int k = 0; k < kv; ++k) {
        const double p = s[k] / L;
        pm = std::max(pm, p);
        for (int e = 0; e < E; ++e) ref[e] += p * from_op(X[((size_t)it * kv + k) * E + e], op);
      }
      peak = std::max(peak, pm);
      for (int e = 0; e < E; ++e) {
        const double got = from_op(u[((size_t)it * R + r) * E + e], op);
        const double err = fabs(got - ref[e]) / (0.05 + fabs(ref[e]));
        worst = std::isfinite(got) ? std::max(worst, err) : 1e30;
      }
    }
  char name[160];
  snprintf(name, sizeof(name), "fold stream items%d kv%d E%d gain%.0f (max row probability %.3f)", items, kv, E, gain, peak);
  // f16 P~ (11 bits) and f16 output against |U| ~ 0.05 .. 1; with peaked rows the f16 rounding of Q' (scores of magnitude 10 .. 60)
  // moves probability between near-tied keys: the same amplification as in tests/test_gpu_headline.py, not a kernel property
  report(name, worst, gain > 1.5f ? 2e-2 : 6e-3);
}

// ------------------------------------------------------------------------------------------------
// weight-gradient GEMM dW = dY^T X (+ bias gradient), row-major and head-major dY, ragged M, accumulate
static void test_gemm_tn(int op, int M, int N, int K, bool headmajor, bool accumulate) {
  const int kv = 37;                           // head-major: rows = items * kv tokens, blocks of 64 cols = "heads"
  const int items = headmajor ? (M + kv - 1) / kv : 1;
  if (headmajor) M = items * kv;
  std::vector<uint16_t> Y((size_t)M * N), X((size_t)M * K);
  std::vector<float> Yf(Y.size()), Xf(X.size());
  for (size_t i = 0; i < Y.size(); ++i) { Y[i] = to_op(frand(), op); Yf[i] = from_op(Y[i], op); }
  for (size_t i = 0; i < X.size(); ++i) { X[i] = to_op(frand(), op); Xf[i] = from_op(X[i], op); }
  // device layout of dY
  std::vector<uint16_t> Yd(Y.size());
  const int nblocks = N / 64;
  if (headmajor) {
    for (int it = 0; it < items; ++it)
      for (int b = 0; b < nblocks; ++b)
        for (int t = 0; t < kv; ++t)
          for (int d = 0; d < 64; ++d) Yd[(((size_t)it * nblocks + b) * kv + t) * 64 + d] = Y[(size_t)(it * kv + t) * N + b * 64 + d];
  } else {
    Yd = Y;
  }
  std::vector<float> W0((size_t)N * K), B0(N);
  for (auto& v : W0) v = frand();
  for (auto& v : B0) v = frand();
  Dev<uint16_t> dY(Yd), dX(X);
  Dev<float> dW(W0), dB(B0), dB2(B0);
  GemmTnArgs a;
  memset(&a, 0, sizeof(a));
  a.dY = dY.p; a.X = dX.p; a.dW = dW.p;
  a.M = M; a.N = N; a.K = K; a.ldw = K; a.accumulate = accumulate;
  a.xv = RowView{0, M, K}; a.x_block_stride = 64;
  if (headmajor) { a.yv = RowView{(long long)nblocks * kv * 64, kv, 64}; a.y_block_stride = (long long)kv * 64; }
  else { a.yv = RowView{0, M, N}; a.y_block_stride = 64; }
  a.db = dB2.p;                                  // fused bias gradient (always accumulates)
  int rc = launch_gemm_tn(a, op, 0);
  rc |= launch_colsum(dY.p, a.y_block_stride, a.yv, M, N, dB.p, accumulate, op, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> W = dW.get(), B = dB.get(), B2 = dB2.get();
  double worst = rc ? 1e30 : 0, worstb = rc ? 1e30 : 0, worstf = worstb;
  for (int n = 0; n < N && !rc; ++n) {
    double bs = accumulate ? B0[n] : 0;
    for (int m = 0; m < M; ++m) bs += Yf[(size_t)m * N + n];
    worstb = std::max(worstb, fabs(bs - B[n]) / (1 + fabs(bs)));
    const double bf = bs + (accumulate ? 0 : B0[n]);
    worstf = std::max(worstf, fabs(bf - B2[n]) / (1 + fabs(bf)));
    for (int k = 0; k < K; ++k) {
      double acc = accumulate ? W0[(size_t)n * K + k] : 0;
      for (int m = 0; m < M; ++m) acc += (double)Yf[(size_t)m * N + n] * Xf[(size_t)m * K + k];
      worst = std::max(worst, fabs(acc - W[(size_t)n * K + k]) / (1 + fabs(acc)));
    }
  }
  char name[128];
  snprintf(name, sizeof(name), "gemm_tn %s M%d N%d K%d %s%s", op == OP_F16 ? "f16" : "bf16", M, N, K, headmajor ? "head-major" : "row-major", accumulate ? " +=" : "");
  report(name, worst, 2e-5);
  report(std::string(name) + " colsum", worstb, 2e-5);
  report(std::string(name) + " fused bias grad", worstf, 2e-5);
}

// ------------------------------------------------------------------------------------------------
// several weight gradients in one launch (launch_gemm_tn_group): jobs of different N / K sharing the contraction length, accumulating into
// pre-filled dW / db; each job against the double-precision product
static void test_gemm_tn_group(int op, int M, int njobs) {
  const int Ns[4] = {128, 64, 192, 64}, Ks[4] = {64, 192, 128, 256};
  std::vector<std::vector<uint16_t>> Y(njobs), X(njobs);
  std::vector<std::vector<float>> W0(njobs), B0(njobs);
  std::vector<Dev<uint16_t>*> dY, dX;
  std::vector<Dev<float>*> dW, dB;
  std::vector<GemmTnArgs> jobs(njobs);
  for (int j = 0; j < njobs; ++j) {
    const int N = Ns[j], K = Ks[j];
    Y[j].resize((size_t)M * N); X[j].resize((size_t)M * K); W0[j].resize((size_t)N * K); B0[j].resize(N);
    for (auto& v : Y[j]) v = to_op(frand(), op);
    for (auto& v : X[j]) v = to_op(frand(), op);
    for (auto& v : W0[j]) v = frand();
    for (auto& v : B0[j]) v = frand();
    dY.push_back(new Dev<uint16_t>(Y[j])); dX.push_back(new Dev<uint16_t>(X[j])); dW.push_back(new Dev<float>(W0[j])); dB.push_back(new Dev<float>(B0[j]));
    GemmTnArgs& a = jobs[j];
    memset(&a, 0, sizeof(a));
    a.dY = dY[j]->p; a.X = dX[j]->p; a.dW = dW[j]->p; a.db = j == 1 ? nullptr : dB[j]->p;
    a.yv = RowView{0, M, N}; a.xv = RowView{0, M, K}; a.y_block_stride = 64; a.x_block_stride = 64;
    a.M = M; a.N = N; a.K = K; a.ldw = K; a.accumulate = 1;
  }
  const int rc = launch_gemm_tn_group(jobs.data(), njobs, op, 0);
  CK(hipDeviceSynchronize());
  double worst = rc ? 1e30 : 0;
  for (int j = 0; j < njobs && !rc; ++j) {
    const int N = Ns[j], K = Ks[j];
    std::vector<float> w = dW[j]->get(), b = dB[j]->get();
    for (int n = 0; n < N; ++n) {
      double bs = B0[j][n];
      for (int m = 0; m < M; ++m) bs += from_op(Y[j][(size_t)m * N + n], op);
      if (j != 1) worst = std::max(worst, fabs(b[n] - bs) / (1 + fabs(bs)));
      else if (b[n] != B0[j][n]) worst = 1e30;                       // no bias gradient asked for: untouched
      for (int k = 0; k < K; ++k) {
        double acc = W0[j][(size_t)n * K + k];
        for (int m = 0; m < M; ++m) acc += (double)from_op(Y[j][(size_t)m * N + n], op) * from_op(X[j][(size_t)m * K + k], op);
        worst = std::max(worst, fabs(w[(size_t)n * K + k] - acc) / (1 + fabs(acc)));
      }
    }
  }
  char name[128];
  snprintf(name, sizeof(name), "gemm_tn group of %d jobs %s M%d", njobs, op == OP_F16 ? "f16" : "bf16", M);
  report(name, worst, 2e-4);
  for (auto p : dY) delete p; for (auto p : dX) delete p; for (auto p : dW) delete p; for (auto p : dB) delete p;
}

static void test_ln_rows(int op) {
  const int items = 5, S = 9, H = 768, rpi = 4, off = 2;
  std::vector<float> x((size_t)items * S * H), g(H), b(H);
  for (auto& v : x) v = frand(2.f) + 0.7f;
  for (auto& v : g) v = 1.f + frand(0.1f);
  for (auto& v : b) v = frand(0.1f);
  Dev<float> dx(x), dg(g), db(b), dy((size_t)items * S * H);
  Dev<uint16_t> dy16((size_t)items * rpi * H);
  dy.fill(0);
  RowView v{(long long)S * H, rpi, H}, c{0, items * rpi, H};
  int rc = launch_ln_rows(dx.p + off * H, v, items * rpi, H, dg.p, db.p, 1e-12f, dy.p + off * H, v, dy16.p, c, op, 0);
  CK(hipDeviceSynchronize());
  double worst = rc ? 1e30 : 0, worst16 = rc ? 1e30 : 0;
  std::vector<float> y = dy.get();
  std::vector<uint16_t> y16 = dy16.get();
  for (int m = 0; m < items * rpi && !rc; ++m) {
    const float* xr = x.data() + off * H + voff(v, m);
    double mu = 0, var = 0;
    for (int i = 0; i < H; ++i) mu += xr[i];
    mu /= H;
    for (int i = 0; i < H; ++i) var += (xr[i] - mu) * (xr[i] - mu);
    var /= H;
    for (int i = 0; i < H; ++i) {
      const double ref = (xr[i] - mu) / sqrt(var + 1e-12) * g[i] + b[i];
      worst = std::max(worst, fabs(ref - y[off * H + voff(v, m) + i]));
      worst16 = std::max(worst16, fabs(ref - from_op(y16[(size_t)m * H + i], op)) / (1 + fabs(ref)));
    }
  }
  report(std::string("ln_rows f32 out ") + (op == OP_F16 ? "f16" : "bf16"), worst, 2e-5);
  report("ln_rows op-dtype copy", worst16, op == OP_F16 ? 1e-3 : 8e-3);
}

static void test_modality_ln(int x_dtype, int E) {
  const int src_items = 4, items = 6, tokens = 5;
  std::vector<float> xf((size_t)src_items * tokens * E), g(E), b(E);
  for (auto& v : xf) v = frand(3.f) - 1.f;
  for (auto& v : g) v = 1.f + frand(0.1f);
  for (auto& v : b) v = frand(0.1f);
  std::vector<long long> idx = {3, 0, 2, 2, 1, 0};
  std::vector<uint16_t> x16(xf.size());
  if (x_dtype != 0)
    for (size_t i = 0; i < xf.size(); ++i) {
      x16[i] = to_op(xf[i], x_dtype == 1 ? OP_F16 : OP_BF16);
      xf[i] = from_op(x16[i], x_dtype == 1 ? OP_F16 : OP_BF16);
    }
  Dev<float> dxf(xf), dg(g), db(b);
  Dev<uint16_t> dx16(x16), dout((size_t)items * tokens * E);
  Dev<long long> didx(idx);
  const void* xp = x_dtype == 0 ? (const void*)dxf.p : (const void*)dx16.p;
  int rc = launch_modality_ln(xp, x_dtype, didx.p, items, tokens, E, dg.p, db.p, 1e-5f, dout.p, OP_F16, 0);
  CK(hipDeviceSynchronize());
  double worst = rc ? 1e30 : 0;
  std::vector<uint16_t> out = dout.get();
  for (int it = 0; it < items && !rc; ++it)
    for (int t = 0; t < tokens; ++t) {
      const float* xr = xf.data() + ((size_t)idx[it] * tokens + t) * E;
      double mu = 0, var = 0;
      for (int i = 0; i < E; ++i) mu += xr[i];
      mu /= E;
      for (int i = 0; i < E; ++i) var += (xr[i] - mu) * (xr[i] - mu);
      var /= E;
      for (int i = 0; i < E; ++i) {
        const double ref = (xr[i] - mu) / sqrt(var + 1e-5) * g[i] + b[i];
        worst = std::max(worst, fabs(ref - from_op(out[((size_t)it * tokens + t) * E + i], OP_F16)) / (1 + fabs(ref)));
      }
    }
  char name[96];
  snprintf(name, sizeof(name), "modality_ln x_dtype%d E%d gather", x_dtype, E);
  report(name, worst, 1e-3);
}

static void test_embed() {
  const int items = 3, L = 5, Q = 32, H = 768, vocab = 50;
  std::vector<float> query((size_t)Q * H), word((size_t)vocab * H), pos((size_t)16 * H), g(H), b(H);
  for (auto& v : query) v = frand(0.02f);
  for (auto& v : word) v = frand(0.02f);
  for (auto& v : pos) v = frand(0.02f);
  for (auto& v : g) v = 1.f + frand(0.1f);
  for (auto& v : b) v = frand(0.1f);
  std::vector<long long> ids = {1, 49, 7, 0, 3, 9, 9, 9, 2, 4, 48, 47, 46, 45, 44};
  Dev<float> dq(query), dw(word), dp(pos), dg(g), db(b), dh((size_t)items * (Q + L) * H);
  Dev<uint16_t> dh16((size_t)items * (Q + L) * H);
  Dev<long long> dids(ids);
  int rc = launch_embed_ln(dids.p, items, L, Q, H, vocab, dq.p, 0, dw.p, dp.p, dg.p, db.p, 1e-12f, dh.p, dh16.p, nullptr, OP_F16, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> h = dh.get();
  double worst = rc ? 1e30 : 0;
  std::vector<double> row(H);
  for (int n = 0; n < items && !rc; ++n)
    for (int s = 0; s < Q + L; ++s) {
      for (int i = 0; i < H; ++i)
        row[i] = s < Q ? query[(size_t)s * H + i] : (double)word[(size_t)ids[n * L + s - Q] * H + i] + pos[(size_t)(s - Q) * H + i];
      double mu = 0, var = 0;
      for (int i = 0; i < H; ++i) mu += row[i];
      mu /= H;
      for (int i = 0; i < H; ++i) var += (row[i] - mu) * (row[i] - mu);
      var /= H;
      for (int i = 0; i < H; ++i)
        worst = std::max(worst, fabs((row[i] - mu) / sqrt(var + 1e-12) * g[i] + b[i] - h[((size_t)n * (Q + L) + s) * H + i]));
    }
  report("embed_ln", worst, 2e-5);
}

static void test_score() {
  const int items = 37, Q = 32, H = 768;
  std::vector<float> z((size_t)items * Q * H), t((size_t)items * H);
  for (auto& v : z) v = frand();
  for (auto& v : t) v = frand();
  Dev<float> dz(z), dt(t), dsim((size_t)items * Q), dlog(items);
  int rc = launch_cosine_score(dz.p, dt.p, items, items, Q, H, 1e-8f, dsim.p, dlog.p, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> sim = dsim.get(), lg = dlog.get();
  double worst = rc ? 1e30 : 0;
  for (int n = 0; n < items && !rc; ++n) {
    double best = -1e9;
    for (int q = 0; q < Q; ++q) {
      double dot = 0, zz = 0, tt = 0;
      for (int i = 0; i < H; ++i) {
        const double a = z[((size_t)n * Q + q) * H + i], b = t[(size_t)n * H + i];
        dot += a * b; zz += a * a; tt += b * b;
      }
      const double s = dot / (sqrt(zz) * sqrt(tt));
      worst = std::max(worst, fabs(s - sim[(size_t)n * Q + q]));
      best = std::max(best, s);
    }
    worst = std::max(worst, fabs(best - lg[n]));
  }
  report("cosine_score", worst, 2e-6);
  // span: bit-exact integers vs the same two-rounding rule on the host
  const int videos = 5, clips = 61;
  std::vector<float> lo((size_t)videos * clips);
  for (auto& v : lo) v = frand();
  lo[3] = lo[17] = 9.f;  // tie: first argmax wins
  Dev<float> dl(lo);
  Dev<int> dsp(videos * 2);
  rc = launch_span(dl.p, videos, clips, 0.5f, dsp.p, 0);
  CK(hipDeviceSynchronize());
  std::vector<int> sp = dsp.get();
  int bad = rc ? 1 : 0;
  for (int v = 0; v < videos && !rc; ++v) {
    const float* x = lo.data() + (size_t)v * clips;
    float hi = x[0], mn = x[0];
    int arg = 0;
    for (int i = 1; i < clips; ++i) { if (x[i] > hi) { hi = x[i]; arg = i; } mn = std::min(mn, x[i]); }
    volatile float prod = 0.5f * (hi - mn);
    volatile float thr = mn + prod;
    int s = arg, e = arg;
    while (s - 1 >= 0 && x[s - 1] >= thr) --s;
    while (e + 1 < clips && x[e + 1] >= thr) ++e;
    if (sp[2 * v] != s || sp[2 * v + 1] != e) ++bad;
  }
  report("span_from_logits (integer mismatches)", bad, 0);
}

// ------------------------------------------------------------------------------------------------
// backward kernels (config 5) against double-precision host loops
static void test_ln_bwd() {
  const int items = 5, S = 9, H = 768, rpi = 4, off = 2, rows = items * rpi;
  std::vector<float> x((size_t)items * S * H), dy((size_t)rows * H), add((size_t)rows * H), g(H);
  for (auto& v : x) v = frand(2.f) + 0.3f;
  for (auto& v : dy) v = frand();
  for (auto& v : add) v = frand();
  for (auto& v : g) v = 1.f + frand(0.2f);
  std::vector<float> dg0(H), db0(H);
  for (auto& v : dg0) v = frand();
  for (auto& v : db0) v = frand();
  Dev<float> dx_(x), ddy(dy), dadd(add), dgam(g), dout((size_t)items * S * H), ddg(dg0), ddb(db0);
  Dev<uint16_t> d16((size_t)rows * H);
  dout.fill(0);
  LnBwdArgs a;
  memset(&a, 0, sizeof(a));
  const RowView v{(long long)S * H, rpi, H}, c{0, rows, H};
  a.dy = ddy.p; a.dyv = c; a.x = dx_.p + off * H; a.xv = v; a.gamma = dgam.p; a.eps = 1e-12f; a.rows = rows;
  a.dx = dout.p + off * H; a.dxv = v; a.add = dadd.p; a.addv = c; a.dx16 = d16.p; a.dx16v = c; a.dgamma = ddg.p; a.dbeta = ddb.p;
  const int rc = launch_ln_bwd(a, H, OP_F16, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> out = dout.get(), dgo = ddg.get(), dbo = ddb.get();
  std::vector<uint16_t> o16 = d16.get();
  std::vector<double> dgr(dg0.begin(), dg0.end()), dbr(db0.begin(), db0.end());
  double worst = rc ? 1e30 : 0, worst16 = worst;
  for (int m = 0; m < rows && !rc; ++m) {
    const float* xr = x.data() + off * H + voff(v, m);
    double mu = 0, var = 0;
    for (int i = 0; i < H; ++i) mu += xr[i];
    mu /= H;
    for (int i = 0; i < H; ++i) var += (xr[i] - mu) * (xr[i] - mu);
    var /= H;
    const double rstd = 1 / sqrt(var + 1e-12);
    double sg = 0, sgx = 0;
    for (int i = 0; i < H; ++i) { const double xh = (xr[i] - mu) * rstd, gg = (double)dy[(size_t)m * H + i] * g[i]; sg += gg; sgx += gg * xh; }
    sg /= H; sgx /= H;
    for (int i = 0; i < H; ++i) {
      const double xh = (xr[i] - mu) * rstd, gg = (double)dy[(size_t)m * H + i] * g[i];
      const double ref = rstd * (gg - sg - xh * sgx) + add[(size_t)m * H + i];
      worst = std::max(worst, fabs(ref - out[off * H + voff(v, m) + i]) / (1 + fabs(ref)));
      worst16 = std::max(worst16, fabs(ref - from_op(o16[(size_t)m * H + i], OP_F16)) / (1 + fabs(ref)));
      dgr[i] += (double)dy[(size_t)m * H + i] * xh;
      dbr[i] += dy[(size_t)m * H + i];
    }
  }
  double worstp = rc ? 1e30 : 0;
  for (int i = 0; i < H; ++i) worstp = std::max({worstp, fabs(dgr[i] - dgo[i]) / (1 + fabs(dgr[i])), fabs(dbr[i] - dbo[i]) / (1 + fabs(dbr[i]))});
  report("ln_bwd dx (+ residual stream)", worst, 2e-5);
  report("ln_bwd dx operand-dtype copy", worst16, 1e-3);
  report("ln_bwd dgamma / dbeta (atomics)", worstp, 2e-5);
}

static void test_gelu_transpose_embed() {
  const long long n = 4096;
  std::vector<uint16_t> u(n), df(n);
  for (auto& v : u) v = to_op(frand(2.f), OP_F16);
  for (auto& v : df) v = to_op(frand(), OP_F16);
  Dev<uint16_t> du(u), ddf(df), df_(n), db_(n);
  int rc = launch_gelu(du.p, nullptr, df_.p, n, 0, OP_F16, 0) | launch_gelu(du.p, ddf.p, db_.p, n, 1, OP_F16, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> f = df_.get(), b = db_.get();
  double w1 = rc ? 1e30 : 0, w2 = w1;
  for (long long i = 0; i < n && !rc; ++i) {
    const double x = from_op(u[i], OP_F16), d = from_op(df[i], OP_F16);
    const double ge = 0.5 * x * (1 + erf(x / sqrt(2.0))), gp = 0.5 * (1 + erf(x / sqrt(2.0))) + x * exp(-0.5 * x * x) / sqrt(2 * M_PI);
    w1 = std::max(w1, fabs(ge - from_op(f[i], OP_F16)) / (1 + fabs(ge)));
    w2 = std::max(w2, fabs(d * gp - from_op(b[i], OP_F16)) / (1 + fabs(d * gp)));
  }
  report("gelu forward (elementwise)", w1, 1e-3);
  report("gelu backward (elementwise)", w2, 1e-3);
  const int R = 70, C = 45;
  std::vector<uint16_t> m((size_t)R * C);
  for (size_t i = 0; i < m.size(); ++i) m[i] = (uint16_t)i;
  Dev<uint16_t> dm(m), dt((size_t)R * C);
  rc = launch_transpose16(dm.p, dt.p, R, C, OP_F16, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> t = dt.get();
  int bad = rc ? 1 : 0;
  for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) bad += t[(size_t)c * R + r] != m[(size_t)r * C + c];
  report("transpose16 (mismatches)", bad, 0);
  // embeddings backward
  const int items = 3, L = 4, Q = 32, H = 256, vocab = 10, S = Q + L;
  std::vector<float> de((size_t)items * S * H);
  for (auto& v : de) v = frand();
  std::vector<long long> ids = {1, 9, 1, 0, 3, 3, 3, 2, 9, 8, 7, 1};
  Dev<float> dde(de), dq((size_t)Q * H), dp((size_t)L * H), dw((size_t)vocab * H);
  dq.fill(0); dp.fill(0); dw.fill(0);
  Dev<long long> dids(ids);
  rc = launch_embed_bwd(dde.p, dids.p, items, L, Q, H, vocab, dq.p, dp.p, dw.p, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> q = dq.get(), pp = dp.get(), w = dw.get();
  std::vector<double> qr((size_t)Q * H, 0), pr((size_t)L * H, 0), wr((size_t)vocab * H, 0);
  for (int n2 = 0; n2 < items; ++n2) for (int s2 = 0; s2 < S; ++s2) for (int c = 0; c < H; ++c) {
    const double g = de[((size_t)n2 * S + s2) * H + c];
    if (s2 < Q) qr[(size_t)s2 * H + c] += g; else { pr[(size_t)(s2 - Q) * H + c] += g; wr[(size_t)ids[n2 * L + s2 - Q] * H + c] += g; }
  }
  double we = rc ? 1e30 : 0;
  for (size_t i = 0; i < qr.size(); ++i) we = std::max(we, fabs(qr[i] - q[i]));
  for (size_t i = 0; i < pr.size(); ++i) we = std::max(we, fabs(pr[i] - pp[i]));
  for (size_t i = 0; i < wr.size(); ++i) we = std::max(we, fabs(wr[i] - w[i]));
  report("embed_bwd (query / position / word grads)", we, 1e-5);
}

// attention backward + the forward's LSE: self (packed QKV, mask) or cross (head-major)
static void test_attn_bwd(int items, int heads, int q_rows, int kv_len, bool self, bool valu = false, int op_ = OP_F16) {
  const int op = op_, H = heads * 64, ldqkv = 3 * H;
  std::vector<uint16_t> Q, K, V, dO((size_t)items * q_rows * H);
  std::vector<long long> mask;
  if (self) {
    Q.resize((size_t)items * q_rows * ldqkv);
    for (auto& v : Q) v = to_op(frand(), op);
    mask.assign((size_t)items * kv_len, 1);
    for (int n = 0; n < items; ++n) for (int t = 32 + (n * 5) % (kv_len - 31); t < kv_len; ++t) mask[(size_t)n * kv_len + t] = 0;
  } else {
    Q.resize((size_t)items * q_rows * H); K.resize((size_t)items * heads * kv_len * 64); V.resize(K.size());
    for (auto& v : Q) v = to_op(frand(), op);
    for (auto& v : K) v = to_op(frand(), op);
    for (auto& v : V) v = to_op(frand(), op);
  }
  for (auto& v : dO) v = to_op(frand(), op);
  Dev<uint16_t> dQ_(Q), dK_(K), dV_(V), ddO(dO), dOut((size_t)items * q_rows * H);
  Dev<long long> dM(mask);
  Dev<float> dlse((size_t)items * heads * q_rows);
  Dev<uint16_t> gQ(self ? Q.size() : Q.size()), gK(self ? 1 : K.size()), gV(self ? 1 : V.size());
  gQ.fill(0);
  AttnArgs f;
  memset(&f, 0, sizeof(f));
  AttnBwdArgs b;
  memset(&b, 0, sizeof(b));
  f.Q = dQ_.p; f.O = dOut.p; f.o_item_stride = (long long)q_rows * H; f.o_ld = H;
  if (self) {
    f.K = dQ_.p + H; f.V = dQ_.p + 2 * H;
    f.q_item_stride = f.k_item_stride = f.v_item_stride = (long long)q_rows * ldqkv;
    f.q_ld = f.k_ld = f.v_ld = ldqkv; f.k_head_stride = f.v_head_stride = 64;
    f.mask = dM.p; f.mask_ld = kv_len;
    b.dQ = gQ.p; b.dK = gQ.p + H; b.dV = gQ.p + 2 * H;
    b.dq_item_stride = b.dk_item_stride = b.dv_item_stride = f.q_item_stride; b.dq_ld = b.dk_ld = b.dv_ld = ldqkv;
    b.dk_head_stride = b.dv_head_stride = 64;
  } else {
    f.K = dK_.p; f.V = dV_.p; f.q_item_stride = (long long)q_rows * H; f.q_ld = H;
    f.k_item_stride = f.v_item_stride = (long long)heads * kv_len * 64; f.k_head_stride = f.v_head_stride = (long long)kv_len * 64;
    f.k_ld = f.v_ld = 64;
    b.dQ = gQ.p; b.dK = gK.p; b.dV = gV.p; b.dq_item_stride = f.q_item_stride; b.dq_ld = H;
    b.dk_item_stride = b.dv_item_stride = f.k_item_stride; b.dk_head_stride = b.dv_head_stride = f.k_head_stride; b.dk_ld = b.dv_ld = 64;
  }
  f.items = items; f.heads = heads; f.q_rows = q_rows; f.kv_len = kv_len; f.scale = 0.125f; f.nsplit = 1; f.lse = dlse.p;
  int rc = launch_attention(f, op, 0);
  b.Q = f.Q; b.K = f.K; b.V = f.V; b.O = dOut.p; b.dO = ddO.p;
  b.q_item_stride = f.q_item_stride; b.o_item_stride = f.o_item_stride; b.q_ld = f.q_ld; b.o_ld = f.o_ld;
  b.k_item_stride = f.k_item_stride; b.k_head_stride = f.k_head_stride; b.v_item_stride = f.v_item_stride; b.v_head_stride = f.v_head_stride;
  b.k_ld = f.k_ld; b.v_ld = f.v_ld; b.mask = f.mask; b.mask_ld = f.mask_ld; b.lse = dlse.p;
  b.items = items; b.heads = heads; b.q_rows = q_rows; b.kv_len = kv_len; b.scale = 0.125f;
#ifdef MRA_GEMM_EXPERIMENTS
  attn_bwd_force_valu(valu ? 1 : 0);
#endif
  rc |= launch_attn_bwd(b, op, 0);
#ifdef MRA_GEMM_EXPERIMENTS
  attn_bwd_force_valu(0);
#endif
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> rQ = gQ.get(), rK = gK.get(), rV = gV.get();
  std::vector<float> lse = dlse.get();
  double wl = rc ? 1e30 : 0, wq = wl, wk = wl, wv = wl;
  std::vector<double> p(kv_len), dp(kv_len);
  for (int n = 0; n < items && !rc; ++n)
    for (int h = 0; h < heads; ++h) {
      std::vector<double> dK((size_t)kv_len * 64, 0), dV((size_t)kv_len * 64, 0);
      auto kp = [&](int t) { return self ? Q.data() + ((size_t)n * q_rows + t) * ldqkv + H + h * 64 : K.data() + (((size_t)n * heads + h) * kv_len + t) * 64; };
      auto vp = [&](int t) { return self ? Q.data() + ((size_t)n * q_rows + t) * ldqkv + 2 * H + h * 64 : V.data() + (((size_t)n * heads + h) * kv_len + t) * 64; };
      for (int q = 0; q < q_rows; ++q) {
        const uint16_t* qp = self ? Q.data() + ((size_t)n * q_rows + q) * ldqkv + h * 64 : Q.data() + ((size_t)n * q_rows + q) * H + h * 64;
        const uint16_t* dop = dO.data() + ((size_t)n * q_rows + q) * H + h * 64;
        double mx = -1e300;
        for (int t = 0; t < kv_len; ++t) {
          double acc = 0;
          for (int d = 0; d < 64; ++d) acc += (double)from_op(qp[d], op) * from_op(kp(t)[d], op);
          acc *= 0.125;
          if (self) acc += (1.0 - (double)mask[(size_t)n * kv_len + t]) * -10000.0;
          p[t] = acc; mx = std::max(mx, acc);
        }
        double den = 0;
        for (int t = 0; t < kv_len; ++t) { p[t] = exp(p[t] - mx); den += p[t]; }
        const double lref = (mx + log(den)) / log(2.0);
        wl = std::max(wl, fabs(lref - lse[((size_t)n * heads + h) * q_rows + q]) / (1 + fabs(lref)));
        double delta = 0;
        for (int t = 0; t < kv_len; ++t) {
          p[t] /= den;
          double acc = 0;
          for (int d = 0; d < 64; ++d) acc += (double)from_op(dop[d], op) * from_op(vp(t)[d], op);
          dp[t] = acc; delta += p[t] * acc;
        }
        for (int d = 0; d < 64; ++d) {
          double dq = 0;
          for (int t = 0; t < kv_len; ++t) dq += p[t] * (dp[t] - delta) * 0.125 * from_op(kp(t)[d], op);
          const size_t qi = self ? ((size_t)n * q_rows + q) * ldqkv + h * 64 + d : ((size_t)n * q_rows + q) * H + h * 64 + d;
          wq = std::max(wq, fabs(dq - from_op(rQ[qi], op)));
        }
        for (int t = 0; t < kv_len; ++t) {
          const double ds = p[t] * (dp[t] - delta) * 0.125;
          for (int d = 0; d < 64; ++d) { dK[(size_t)t * 64 + d] += ds * from_op(qp[d], op); dV[(size_t)t * 64 + d] += p[t] * from_op(dop[d], op); }
        }
      }
      for (int t = 0; t < kv_len; ++t) for (int d = 0; d < 64; ++d) {
        const size_t ki = self ? ((size_t)n * q_rows + t) * ldqkv + H + h * 64 + d : (((size_t)n * heads + h) * kv_len + t) * 64 + d;
        const size_t vi = self ? ki + H : ki;
        wk = std::max(wk, fabs(dK[(size_t)t * 64 + d] - from_op(self ? rQ[ki] : rK[ki], op)));
        wv = std::max(wv, fabs(dV[(size_t)t * 64 + d] - from_op(self ? rQ[vi] : rV[vi], op)));
      }
    }
  char name[96];
  snprintf(name, sizeof(name), "attn %s %s%s items%d heads%d q%d kv%d", self ? "self" : "cross", op == OP_F16 ? "f16" : "bf16",
           valu ? " valu" : " mfma", items, heads, q_rows, kv_len);
  const double tol = op == OP_F16 ? 6e-3 : 4e-2;
  report(std::string(name) + " forward LSE", wl, op == OP_F16 ? 2e-3 : 1e-2);
  report(std::string(name) + " backward dQ", wq, tol);
  report(std::string(name) + " backward dK", wk, tol);
  report(std::string(name) + " backward dV", wv, tol);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  int dev = 0;
  CK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);

  test_gemm_tn(OP_F16, 100, 64, 64, false, false);
  test_gemm_tn(OP_F16, 1040, 128, 192, false, true);
  test_gemm_tn(OP_BF16, 333, 192, 128, false, false);
  test_gemm_tn(OP_F16, 300, 128, 64, true, true);
  test_gemm_tn_group(OP_F16, 1040, 4);
  test_gemm_tn_group(OP_BF16, 333, 3);
  test_gemm_tn_group(OP_F16, 100, 2);
  test_ln_bwd();
  test_gelu_transpose_embed();
  test_attn_bwd(3, 2, 45, 45, true);
  test_attn_bwd(2, 2, 160, 160, true);
  test_attn_bwd(3, 2, 32, 257, false);
  test_attn_bwd(2, 3, 64, 64, true);
  test_attn_bwd(2, 2, 33, 33, true);
  test_attn_bwd(1, 12, 32, 1000, false);
  test_attn_bwd(2, 2, 41, 41, true, false, OP_BF16);
  test_attn_bwd(2, 2, 32, 100, false, false, OP_BF16);
#ifdef MRA_GEMM_EXPERIMENTS
  test_attn_bwd(3, 2, 45, 45, true, true);      // the fp32 VALU kernel (A/B switch, experiment library only)
  test_attn_bwd(3, 2, 32, 257, false, true);
#endif
  test_ln_rows(OP_F16);
  test_ln_rows(OP_BF16);
  test_modality_ln(0, 1408);
  test_modality_ln(1, 768);
  test_modality_ln(2, 1408);
  test_modality_ln(0, 2048);
  test_embed();
  test_score();

  // GEMM: every tile config x epilogue, ragged M, row views, grouped
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int t = cfg == 0 ? 64 : (cfg == 1 ? 128 : 256);
    test_gemm(cfg, EPI_OP, OP_F16, 2 * t + 37, 2 * t, 192, true);
    test_gemm(cfg, EPI_GELU_OP, OP_F16, t - 5, t, 64, false, 2);
    test_gemm(cfg, EPI_RES_F32, OP_F16, 3 * t + 1, t, 128, true, 2);
    test_gemm(cfg, EPI_F32, OP_F16, t, 2 * t, 256, false);
    test_gemm(cfg, EPI_KV, OP_F16, 2 * t + 10, 2 * t >= 256 ? 2 * t : 256, 128, false);
    test_gemm(cfg, EPI_OP, OP_BF16, t + 3, t, 128, true);
  }
  test_gemm(-1, EPI_OP, OP_F16, 300, 768, 1408, false);  // automatic config
  // the eight-phase 256 x 256 kernel (even K / 64): every epilogue, ragged M, row views, two problems, K = 128 (one pair) .. 1408
  test_gemm(2, EPI_OP, OP_F16, 2 * 256 + 37, 512, 128, true);
  test_gemm(2, EPI_KV, OP_F16, 2100, 1536, 1408, false);
  test_gemm(2, EPI_GELU_OP, OP_F16, 700, 256, 1408, true);
  test_gemm(2, EPI_RES_F32, OP_F16, 2 * 256 + 37, 512, 384, true, 2);
  test_gemm(2, EPI_F32, OP_BF16, 300, 512, 256, false);
  test_gemm(2, EPI_RES_OP, OP_F16, 600, 512, 640, true);
  test_gemm(2, EPI_OP, OP_BF16, 1000, 768, 768, false, 2);
  test_gemm_masked(2, EPI_RES_F32, OP_F16, 700, 352, 128);
  test_gemm_masked(2, EPI_F32, OP_F16, 300, 1408, 256);
  test_gemm(6, EPI_RES_F32, OP_F16, 1300, 128, 384, true);        // the 128 x 512 form (tile_cfg 7): one column tile, ragged M, views
  test_gemm(6, EPI_F32, OP_BF16, 1000, 128, 128, false);
  test_gemm(6, EPI_RES_OP, OP_F16, 700, 128, 1408, true);
  test_gemm(6, EPI_RES_F32, OP_F16, 512, 128, 6144, false);
  g_test_persist = 1;                                              // the eight-phase kernel as one persistent workgroup per CU: more tiles than CUs, ragged M, views
  test_gemm(2, EPI_OP, OP_F16, 17 * 256 - 90, 4096, 256, true);
  test_gemm(2, EPI_GELU_OP, OP_BF16, 4200, 4096, 128, false);
  test_gemm(2, EPI_KV, OP_F16, 9 * 256 + 10, 8192, 128, false);
  g_test_persist = 0;
  test_gemm_ln_fold(OP_F16, 700, 384, 256, 512, false, 0.f);      // LayerNorm folded into the producer / consumer GEMMs: mixed tiles (N = 256 k + 128), ragged M
  test_gemm_ln_fold(OP_F16, 1300, 1408, 128, 256, true, 0.f);      // the ViT's width: 11 groups, odd number of row tiles, GELU consumer
  test_gemm_ln_fold(OP_F16, 520, 640, 384, 256, false, 8.f);       // row means 8 sigma from zero
  test_gemm_ln_fold(OP_BF16, 600, 512, 128, 512, true, 0.f);       // N = 256 k: full tiles only (tile_cfg 3)
  test_gemm_res_op_stat(OP_F16, 1300, 1408, 128, 0.f);             // the same for the residual stream in the operand dtype: mixed tiles, ragged M
  test_gemm_res_op_stat(OP_F16, 520, 640, 384, 8.f);
  test_gemm_res_op_stat(OP_BF16, 600, 512, 128, 0.f);              // full tiles only
  test_gemm(7, EPI_RES_F32, OP_F16, 1300, 384, 256, true);         // tile_cfg 8: 256-wide tiles + 128 x 512 tail tiles in one launch; odd number of row tiles
  test_gemm(7, EPI_F32, OP_BF16, 1024, 640, 128, false);
  test_gemm(7, EPI_RES_OP, OP_F16, 2100, 1408, 384, true);
  test_gemm(7, EPI_OP, OP_F16, 1300, 384, 256, true);             // 16-bit outputs through the tail tile's LDS-staged epilogue (gemm_bench qkvpad: un-padded ViT QKV)
  test_gemm(7, EPI_OP, OP_BF16, 1024, 640, 128, false);
  test_gemm(7, EPI_RES_F32, OP_F16, 700, 1408, 1408, false);
  // the ring kernel's exact-fit tiles (tile_cfg 9 / 10 / 11): every epilogue, ragged M, row views, two problems, 1 .. 48 K steps (fewer than,
  // exactly, and many more than the ring holds)
  test_gemm(8, EPI_OP, OP_F16, 2 * 128 + 37, 288, 192, true);
  test_gemm(8, EPI_OP, OP_F16, 700, 144, 768, false);
  test_gemm(8, EPI_GELU_OP, OP_F16, 100, 288, 256, false);
  test_gemm(8, EPI_OP, OP_BF16, 300, 144, 128, false, 2);
  test_gemm(8, EPI_OP, OP_F16, 128 + 5, 432, 64, true, 2);
  test_gemm(9, EPI_GELU_OP, OP_F16, 2 * 128 + 37, 384, 768, true, 2);
  test_gemm(9, EPI_OP, OP_BF16, 128, 192, 64, false);
  test_gemm(9, EPI_OP, OP_F16, 500, 576, 128, true);
  test_gemm(10, EPI_RES_F32, OP_F16, 3 * 64 + 11, 192, 3072, true, 2);
  test_gemm(10, EPI_RES_F32, OP_F16, 500, 768, 768, false);
  test_gemm(10, EPI_OP, OP_F16, 200, 96, 448, true);
  test_gemm(10, EPI_F32, OP_BF16, 64, 288, 64, false);
  test_gemm(10, EPI_GELU_OP, OP_F16, 70, 96, 384, false);
  test_gemm_res_ln(OP_F16, 2048, 768, 768, 1);
  test_gemm_res_ln(OP_F16, 1024 - 30, 768, 3072, 2);
  test_gemm_res_ln(OP_BF16, 200, 768, 128, 1);
  test_gemm_res_ln(OP_F16, 333, 768, 256, 2);
  test_gemm_res_ln(OP_F16, 70, 768, 64, 1);
  // an odd number of K steps takes the loader-wave 256 x 256 kernel (8 compute + 4 loader waves)
  test_gemm(2, EPI_GELU_OP, OP_F16, 600, 512, 1344, false);   // GELU epilogue, pre-activations out to |x| ~ 6
  test_gemm(2, EPI_RES_OP, OP_F16, 2 * 256 + 37, 512, 192, true);
  test_gemm(2, EPI_RES_F32, OP_F16, 2 * 256 + 37, 512, 320, true, 2);
  test_gemm(2, EPI_KV, OP_F16, 256 + 10, 512, 64, false);
  test_gemm(2, EPI_OP, OP_BF16, 300, 512, 192, false);
  test_gemm(2, EPI_F32, OP_F16, 300, 256, 704, false);
  g_test_order = 8;   // column-fastest panels (the ViT's N = 1408 GEMMs)
  test_gemm(2, EPI_RES_F32, OP_F16, 1300, 512, 320, true, 2);
  test_gemm(2, EPI_RES_F32, OP_F16, 1300, 512, 384, true, 2);
  test_gemm_masked(2, EPI_RES_F32, OP_F16, 1500, 1408, 192);
  test_gemm_masked(2, EPI_RES_F32, OP_F16, 1500, 1408, 256);
  test_gemm(1, EPI_OP, OP_F16, 700, 640, 128, true);
  g_test_order = 0;
  test_gemm_masked(2, EPI_RES_F32, OP_F16, 700, 352, 128);   // N = 1.4 column tiles of 256, the tail is neither read nor stored
  test_gemm_masked(2, EPI_F32, OP_F16, 300, 1408, 192);
  test_gemm_masked(1, EPI_RES_F32, OP_BF16, 200, 200, 64);
  test_gemm_batched(1, EPI_F32, OP_F16, 96, 300, 192, 3, true);     // scores: M = 3 x 32 rows, N = kv ragged
  test_gemm_batched(1, EPI_OP, OP_F16, 96, 256, 320, 3, false);     // P . enc^T
  test_gemm_batched(0, EPI_OP, OP_F16, 70, 128, 64, 4, false);      // per-head projections, K = 64
  test_gemm_batched(0, EPI_F32, OP_BF16, 33, 100, 128, 2, true);
  test_gemm_batched(2, EPI_F32, OP_F16, 300, 700, 128, 2, true);
  test_gemm_batched(3, EPI_F32, OP_F16, 384, 300, 192, 2, true);    // the 128 x 384 loader-wave tile
  test_gemm_batched(3, EPI_OP, OP_F16, 384, 256, 320, 3, false);
  test_gemm_batched(3, EPI_OP, OP_BF16, 200, 128, 128, 2, false);   // fewer rows than the tile
  test_gemm_batched(3, EPI_F32, OP_F16, 500, 129, 64, 1, true);     // two row tiles, one K step
  test_gemm_batched(4, EPI_OP, OP_F16, 384, 352, 320, 3, false);    // 176 x 384, compute waves in one column
  test_gemm_batched(4, EPI_OP, OP_BF16, 300, 176, 128, 2, false);
  test_gemm_batched(4, EPI_OP, OP_F16, 500, 528, 64, 1, false);     // two row tiles, one K step
  test_gemm_batched(4, EPI_F32, OP_F16, 384, 300, 192, 2, true);    // scores: ragged N on the 176-row tile
  test_gemm_batched(4, EPI_F32, OP_F16, 384, 300, 384, 2, true, true);   // split-precision scores: A = (hi | lo), the weight slab walked twice
  test_gemm_batched(4, EPI_F32, OP_F16, 384, 530, 2816, 1, true, true);  // ... at the video width (E = 1408)
  test_gemm_batched(3, EPI_F32, OP_BF16, 300, 256, 256, 2, false, true);
  test_gemm_kmajor(OP_F16, 384, 352, 320, 300, 2);                  // P . enc with the weights K-major (transposed LDS reads)
  test_gemm_kmajor(OP_F16, 200, 176, 64, 64, 3);
  test_gemm_kmajor(OP_BF16, 384, 528, 192, 150, 1);
  test_gemm_softpart(OP_F16, 384, 1000, 192, 4);
  test_gemm_softpart(OP_F16, 384, 530, 64, 3);
  test_gemm_softpart(OP_BF16, 384, 400, 128, 2);
  test_gemm_softpart(OP_F16, 200, 1000, 320, 2);                   // M < 384: clamped rows
  test_gemm_pscale(OP_F16, 384, 352, 1000, 2);                      // 6 score tiles, K 1152: the factor ring wraps, NaN tail
  test_gemm_pscale(OP_F16, 384, 176, 150, 3);
  test_gemm_pscale(OP_BF16, 200, 176, 700, 1);
  test_gemm_batched(4, EPI_F32, OP_F16, 100, 177, 64, 2, true);
  test_fold_stream(2, 300, 704, 1.f);       // two score tiles, ragged kv, two E slabs
  test_fold_stream(1, 2100, 704, 1.f);      // 12 score tiles, K loop of 34 steps (not a multiple of the set rotation)
  test_fold_stream(2, 700, 704, 12.f);      // peaked rows: tile factors spread over many powers of two
  test_fold_stream(1, 1000, 704, 60.f);     // near one-hot rows: most tile factors flush to zero
  test_fold_stream(3, 176, 1408, 2.f);      // exactly one tile, the video width
#ifdef MRA_GEMM_EXPERIMENTS   // kernel_check_exp (links tests/native/libmra_hip_exp.so): the A/B main loops of gemm_experiments.inc
  gemm_set_eight_phase(0);                               // the loader-wave kernel on an even number of K steps
  test_gemm(2, EPI_GELU_OP, OP_F16, 600, 512, 1408, false);
  test_gemm(2, EPI_KV, OP_F16, 2100, 1536, 1408, false);
  gemm_set_eight_phase(1);
  gemm_force_variant(1);                                 // the two-buffer main loop kept for A/B runs
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int t = cfg == 0 ? 64 : (cfg == 1 ? 128 : 256);
    test_gemm(cfg, EPI_RES_F32, OP_F16, 2 * t + 37, 2 * t, 192, true, 2);
    test_gemm(cfg, EPI_KV, OP_F16, t + 10, 2 * t >= 256 ? 2 * t : 256, 64, false);
  }
  gemm_force_variant(2);                                 // v1 + L2 prefetch
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int t = cfg == 0 ? 64 : (cfg == 1 ? 128 : 256);
    test_gemm(cfg, EPI_RES_F32, OP_F16, 2 * t + 37, 2 * t, 320, true, 2);
    test_gemm(cfg, EPI_KV, OP_F16, t + 10, 2 * t >= 256 ? 2 * t : 256, 64, false);
  }
  gemm_force_variant(3);                                 // v1 with the LDS-DMA issue spread over the MFMAs
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int t = cfg == 0 ? 64 : (cfg == 1 ? 128 : 256);
    test_gemm(cfg, EPI_RES_F32, OP_F16, 2 * t + 37, 2 * t, 320, true, 2);
    test_gemm(cfg, EPI_KV, OP_F16, t + 10, 2 * t >= 256 ? 2 * t : 256, 64, false);
  }
  gemm_force_variant(5);                                 // warp-specialised 256x256 loop (8 compute + 4 loader waves)
  test_gemm(2, EPI_RES_F32, OP_F16, 2 * 256 + 37, 512, 320, true, 2);
  test_gemm(2, EPI_KV, OP_F16, 256 + 10, 512, 64, false);
  test_gemm(2, EPI_GELU_OP, OP_F16, 700, 256, 1408, true);
  test_gemm(2, EPI_OP, OP_BF16, 300, 512, 192, false);
  {
    unsigned long long* dbg;
    CK(hipMalloc((void**)&dbg, 64));
    CK(hipMemset(dbg, 0, 64));
    gemm_set_debug_buffer(dbg);
    gemm_force_variant(7);                               // ws2: LDS-flag hand-off, no barrier in the K loop
    test_gemm(2, EPI_RES_F32, OP_F16, 2 * 256 + 37, 512, 320, true, 2);
    test_gemm(2, EPI_KV, OP_F16, 256 + 10, 512, 64, false);
    test_gemm(2, EPI_GELU_OP, OP_F16, 700, 256, 1408, true);
    test_gemm(2, EPI_OP, OP_BF16, 300, 512, 192, false);
    test_gemm(2, EPI_KV, OP_F16, 2100, 1536, 1408, false);
    unsigned long long flag = 0;
    CK(hipMemcpy(&flag, dbg, 8, hipMemcpyDeviceToHost));
    report("ws2 spin give-ups (must be 0)", (double)flag, 0);
    gemm_set_debug_buffer(nullptr);
    (void)hipFree(dbg);
  }
  gemm_force_variant(9);                                 // rot: deferred half tile across the barrier
  test_gemm(2, EPI_RES_F32, OP_F16, 2 * 256 + 37, 512, 320, true, 2);
  test_gemm(2, EPI_KV, OP_F16, 256 + 10, 512, 64, false);
  test_gemm(2, EPI_GELU_OP, OP_F16, 700, 256, 1408, true);
  test_gemm(2, EPI_OP, OP_BF16, 300, 512, 128, false);
  gemm_force_variant(8);                                 // 128-deep K steps for the 64 / 128 tiles
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int t = cfg == 0 ? 64 : 128;
    test_gemm(cfg, EPI_OP, OP_F16, 2 * t + 37, 2 * t, 384, true);
    test_gemm(cfg, EPI_GELU_OP, OP_F16, t - 5, t, 128, false, 2);
    test_gemm(cfg, EPI_RES_F32, OP_F16, 3 * t + 1, t, 768, true, 2);
    test_gemm(cfg, EPI_KV, OP_F16, 2 * t + 10, 256, 1408, false);
    test_gemm(cfg, EPI_OP, OP_BF16, t + 3, t, 256, true);
    test_gemm(cfg, EPI_F32, OP_F16, t, 2 * t, 192, false);   // K % 128 != 0: falls back to the 64-deep loop
  }
  gemm_force_variant(0);                                 // ring loop
  test_gemm(2, EPI_OP, OP_F16, 300, 512, 64, false);     // K = 64: fewer slots than the ring holds
  test_gemm(0, EPI_OP, OP_F16, 100, 128, 3072, false);   // long K on the 8-slot ring
  gemm_force_variant(5);
#endif

  // attention
  test_attention(OP_F16, 5, 3, 45, 45, true, 1);     // self, ragged masks, 2 query blocks (13 live rows in the 2nd)
  test_attention(OP_F16, 3, 2, 160, 160, true, 1);   // self, max S
  test_attention(OP_F16, 3, 2, 32, 100, false, 1);   // cross, short kv: one wave per unit
  test_attention(OP_F16, 3, 2, 32, 257, false, 1);   // cross, reference shape: 4 waves per unit
  test_attention(OP_F16, 2, 2, 32, 2100, false, 3);  // cross, grid split + combine
  test_attention(OP_F16, 2, 2, 32, 2100, false, 1, true);  // rescale path forced
  test_attention(OP_BF16, 2, 2, 32, 257, false, 1);
  test_attention(OP_BF16, 2, 2, 64, 64, true, 1);
  if (!quick) {
    test_attention(OP_F16, 2, 12, 32, 8224, false, 4);
    test_gemm(2, EPI_KV, OP_F16, 2100, 1536, 1408, false);
  }
  printf("%d case(s) failed\n", g_fail);
  return g_fail;
}
