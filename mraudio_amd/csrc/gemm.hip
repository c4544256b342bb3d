// MFMA GEMM for every projection on the Q-Former path:  C[m][n] = sum_k A[m][k] * W[n][k] (+ epilogue).
//
// Replaces the nn.Linear calls inside the LAVIS Q-Former that the reference invokes at
// models/xinstructblip.py:286-293 (self/cross Q,K,V and output projections, both feed-forwards)
// and the llm_proj at models/xinstructblip.py:303.  Both operands are K-contiguous ([rows][K]),
// so the product is computed "swapped": the weight rows ride the MFMA row index and the
// activation rows the MFMA column index (D[n][m]); each lane then owns 4 consecutive n of one
// output row m, i.e. 8 contiguous bytes (f16) / 16 bytes (f32) of C.
//
// Structure (gfx950): TN x TM x 64 tiles; operands staged by LDS-DMA (global_load_lds, 16 B per
// lane, 1 KiB per wave instruction) into two LDS buffers; rows are 128 B so ds_read_b128 fragment
// reads would be 8-way bank conflicted: the 16-byte chunk index is XOR-swizzled with
// (row >> 1) & 7, applied on the per-lane SOURCE address (LDS-DMA writes linearly) and on the
// read address.  MFMA = v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulation.
// Workgroups walk tiles in 8-row-tile panels after an XCD-contiguous remap so that the 32
// concurrently running workgroups of one XCD share operand panels in that XCD's L2.
#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

constexpr int BK = 64;         // k elements per tile (128 B rows)
constexpr int ROWB = BK * 2;   // bytes per staged row

__device__ __forceinline__ long long view_off(const RowView& v, int m) {
  int item = m / v.rpi;
  int r = m - item * v.rpi;
  return (long long)item * v.item_stride + (long long)r * v.ld;
}

template <typename T, int TN, int TM, int WGN, int WGM, int EPI>
__global__ void __launch_bounds__(WGN* WGM * 64) gemm_kernel(const GemmArgs args) {
  constexpr int NT = WGN * WGM * 64;
  constexpr int WTN = TN / WGN, WTM = TM / WGM;
  constexpr int FN = WTN / 16, FM = WTM / 16;
  constexpr int IW = TN * 8 / NT, IX = TM * 8 / NT;  // 16-byte chunks per thread per operand tile
  static_assert(TN * 8 % NT == 0 && TM * 8 % NT == 0, "tile/threads mismatch");
  constexpr int BUF = (TN + TM) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn0 = (wave / WGM) * WTN;
  const int wm0 = (wave % WGM) * WTM;

  // ---- tile id: XCD-contiguous remap (bijective for any grid size), then group lookup ----
  int id = blockIdx.x;
  {
    const int nwg = args.total_tiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int g = (args.ngroups > 1 && id >= args.p[1].tile_begin) ? 1 : 0;
  const GemmProb& P = args.p[g];
  int tn, tm;
  {
    const int pid = id - P.tile_begin;
    constexpr int GM = 8;
    const int per_panel = GM * P.ntiles;
    const int panel = pid / per_panel;
    const int first_m = panel * GM;
    const int gsz = min(GM, P.mtiles - first_m);
    const int in_panel = pid - panel * per_panel;
    tm = first_m + in_panel % gsz;
    tn = in_panel / gsz;
  }
  const int n0 = tn * TN, m0 = tm * TM;
  const int K = P.K, M = P.M;

  // ---- per-lane source pointers (swizzled chunk) ----
  const char* srcW[IW];
  const char* srcX[IX];
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
    srcW[i] = (const char*)P.W + ((long long)(n0 + row) * K + c * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
    const int m = min(m0 + row, M - 1);  // rows past M are computed on a clamped row and never stored
    srcX[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
  }
  const int wave_q0 = wave * 64;  // this wave's first chunk inside each i-slab

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * BUF;
    const long long koff = (long long)kt * ROWB;
#pragma unroll
    for (int i = 0; i < IW; ++i) glds16(srcW[i] + koff, base + (wave_q0 + i * NT) * 16);
#pragma unroll
    for (int i = 0; i < IX; ++i) glds16(srcX[i] + koff, base + TN * ROWB + (wave_q0 + i * NT) * 16);
  };

  // fragment read offsets: row (lane & 15), logical chunk 4*ks + (lane >> 4), swizzled
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int r = lane & 15;
    const int c = (4 * ks + (lane >> 4)) ^ ((r >> 1) & 7);
    foff[ks] = r * ROWB + c * 16;
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* wb = smem + (kt & 1) * BUF + wn0 * ROWB;
    const char* xb = smem + (kt & 1) * BUF + TN * ROWB + wm0 * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Vec8<T>::type a[FN], b[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) a[i] = lds_read8<T>(wb + i * 16 * ROWB + foff[ks]);
#pragma unroll
      for (int j = 0; j < FM; ++j) b[j] = lds_read8<T>(xb + j * 16 * ROWB + foff[ks]);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<T>(a[i], b[j], acc[i][j]);
    }
  }

  // ---- epilogue: lane owns C[m][n .. n+3], m = col (lane & 15), n = 4 * (lane >> 4) + reg ----
  const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm0 + j * 16 + lm;
    if (m >= M) continue;
    long long coff = 0, roff = 0;
    int item = 0, tok = 0;
    if constexpr (EPI == EPI_KV) {
      item = m / P.kv_tokens;
      tok = m - item * P.kv_tokens;
    } else {
      coff = view_off(P.c, m);
      if constexpr (EPI == EPI_RES_F32) roff = view_off(P.r, m);
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn0 + i * 16 + ln;
      f32x4 v = acc[i][j];
      if (P.bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(P.bias + n);
        v += bv;
      }
      if constexpr (EPI == EPI_GELU_OP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if constexpr (EPI == EPI_RES_F32) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(P.R + roff + n);
        v += rv;
      }
      if constexpr (EPI == EPI_RES_F32 || EPI == EPI_F32) {
        *reinterpret_cast<f32x4*>((float*)P.C + coff + n) = v;
      } else {
        typename Vec4<T>::type o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
        if constexpr (EPI == EPI_KV) {
          const int hidden = P.kv_heads * 64;
          const int sel = n / hidden;           // cl * 2 + kv
          const int within = n - sel * hidden;  // head * 64 + d
          const int head = within >> 6, d = within & 63;
          const long long dst =
              ((((long long)sel * P.kv_items + item) * P.kv_heads + head) * P.kv_tokens + tok) * 64 + d;
          *reinterpret_cast<typename Vec4<T>::type*>((T*)P.C + dst) = o;
        } else {
          *reinterpret_cast<typename Vec4<T>::type*>((T*)P.C + coff + n) = o;
        }
      }
    }
  }
}

int g_force_cfg = -1;

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_cfg(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr int NT = WGN * WGM * 64;
  constexpr size_t lds = 2 * (TN + TM) * ROWB;
  dim3 grid(a.total_tiles), block(NT);
#define MRA_GEMM_CASE(E)                                                                                  \
  case E: {                                                                                               \
    auto kfn = gemm_kernel<T, TN, TM, WGN, WGM, E>;                                                       \
    if (lds > 64 * 1024) {                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                         (int)lds);                                                       \
      if (e != hipSuccess) return -3;                                                                     \
    }                                                                                                     \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, a);                                                 \
    break;                                                                                                \
  }
  switch (epi) {
    MRA_GEMM_CASE(EPI_OP)
    MRA_GEMM_CASE(EPI_GELU_OP)
    MRA_GEMM_CASE(EPI_RES_F32)
    MRA_GEMM_CASE(EPI_F32)
    MRA_GEMM_CASE(EPI_KV)
    default:
      return -2;
  }
#undef MRA_GEMM_CASE
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

constexpr int kTile[3] = {64, 128, 256};

}  // namespace

void gemm_force_config(int cfg) { g_force_cfg = cfg; }

int gemm_pick_config(const GemmProb* probs, int ngroups) {
  if (g_force_cfg >= 0) return g_force_cfg;
  // Largest tile that still gives every CU work: 256 CUs; aim for >= 2 waves of workgroups
  // with the 64/128 tiles and >= 1 full wave with the 256 tile.
  int best = 0;
  for (int c = 2; c >= 0; --c) {
    const int t = kTile[c];
    long long tiles = 0;
    bool ok = true;
    for (int g = 0; g < ngroups; ++g) {
      if (probs[g].N % t) ok = false;
      tiles += (long long)((probs[g].M + t - 1) / t) * (probs[g].N / t);
    }
    if (!ok) continue;
    const long long need = c == 2 ? 512 : (c == 1 ? 384 : 0);
    if (tiles >= need) {
      best = c;
      break;
    }
  }
  return best;
}

int launch_gemm(const GemmProb* probs, int ngroups, int epi, int op_dtype, hipStream_t stream) {
  if (ngroups < 1 || ngroups > 2) return -1;
  const int cfg = gemm_pick_config(probs, ngroups);
  const int t = kTile[cfg];
  GemmArgs a;
  a.ngroups = ngroups;
  int tiles = 0;
  for (int g = 0; g < ngroups; ++g) {
    a.p[g] = probs[g];
    GemmProb& p = a.p[g];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
    if (p.K % BK || p.N % t) return -1;
    if (p.a.rpi <= 0 || (epi != EPI_KV && p.c.rpi <= 0)) return -1;
    if (epi == EPI_RES_F32 && (!p.R || p.r.rpi <= 0)) return -1;
    if (epi == EPI_KV && (p.kv_tokens <= 0 || p.kv_heads <= 0 || p.kv_items <= 0)) return -1;
    p.mtiles = (p.M + t - 1) / t;
    p.ntiles = p.N / t;
    p.tile_begin = tiles;
    tiles += p.mtiles * p.ntiles;
  }
  if (ngroups == 1) a.p[1] = a.p[0];
  a.total_tiles = tiles;
  if (op_dtype == OP_F16) {
    if (cfg == 2) return launch_cfg<f16, 256, 256, 2, 4>(a, epi, stream);
    if (cfg == 1) return launch_cfg<f16, 128, 128, 2, 2>(a, epi, stream);
    return launch_cfg<f16, 64, 64, 2, 2>(a, epi, stream);
  } else {
    if (cfg == 2) return launch_cfg<bf16, 256, 256, 2, 4>(a, epi, stream);
    if (cfg == 1) return launch_cfg<bf16, 128, 128, 2, 2>(a, epi, stream);
    return launch_cfg<bf16, 64, 64, 2, 2>(a, epi, stream);
  }
}

}  // namespace mra
