// MFMA GEMM for every projection on the Q-Former path:  C[m][n] = sum_k A[m][k] * W[n][k] (+ epilogue).
//
// Replaces the nn.Linear calls inside the LAVIS Q-Former that the reference invokes at
// models/xinstructblip.py:286-293 (self/cross Q,K,V and output projections, both feed-forwards)
// and the llm_proj at models/xinstructblip.py:303.  Both operands are K-contiguous ([rows][K]),
// so the product is computed "swapped": the weight rows ride the MFMA row index and the
// activation rows the MFMA column index (D[n][m]); each lane then owns 4 consecutive n of one
// output row m.  gfx950, v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulation.
//
// Main loops (one per regime):
//   gemm_p8_kernel    256x256x64 in eight phases per pair of K tiles (8 waves, SIMD partners one barrier apart, half-tile DMA
//                     staging with counted vmcnt): the K/V projection, the ViT's four GEMMs, every launch with >= 512 tiles
//                     and an even number of K steps.
//   gemm_ws_kernel    8 compute waves + 4 LDS-DMA loader waves per workgroup: the 176x384 / 128x384 tiles of the folded
//                     cross-attention, and 256x256 with an odd number of K steps.
//   gemm_kernel       64x64 / 128x128 tiles of the 12-layer chain: two 64-deep LDS buffers.
//   gemm_k128_kernel  the same with 128-deep steps, for the long-K down-projections.
// Operands are staged by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction); the
// 16-byte chunk index is XOR-swizzled on the per-lane SOURCE address (LDS-DMA writes linearly) and on
// the ds_read_b128 address, conflict-free for all four 16-lane groups.  16-bit outputs leave through
// LDS as whole 128-byte lines.  Workgroups walk tiles in 8-row-tile panels after an XCD-contiguous
// remap so that the 32 concurrently running workgroups of one XCD share operand panels in its L2.
// Other main loops that were tried and measured (ring, prefetch, flag hand-off ...) live in
// tests/native/gemm_experiments.inc, outside the product tree and the shipped library; verdicts in DESIGN.md section 8.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

__device__ __forceinline__ long long view_off(const RowView& v, int m) {
  int item = m / v.rpi;
  int r = m - item * v.rpi;
  return (long long)item * v.item_stride + (long long)r * v.ld;
}

// XCD-contiguous remap (bijective for any grid size), group lookup, 8-row-tile panels
template <int TN, int TM>
__device__ __forceinline__ GemmProb tile_of(const GemmArgs& args, int id, int& n0, int& m0);

template <int TN, int TM>
__device__ __forceinline__ GemmProb pick_tile(const GemmArgs& args, int& n0, int& m0) {
  int id = blockIdx.x;
  {
    const int nwg = args.total_tiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  return tile_of<TN, TM>(args, id, n0, m0);
}

// tile `id` of the launch (after the XCD remap): its problem (batch entry applied) and origin
template <int TN, int TM>
__device__ __forceinline__ GemmProb tile_of(const GemmArgs& args, int id, int& n0, int& m0) {
  int g = 0;   // the last problem whose first tile is <= id (tile_begin ascends)
#pragma unroll
  for (int k = 1; k < GEMM_MAX_GROUPS; ++k)
    if (k < args.ngroups && id >= args.p[k].tile_begin) g = k;
  GemmProb P = args.p[g];
  int pid = id - P.tile_begin;
  if (P.batch > 1) {   // batch entry first, then the panel order inside it
    const int per = P.mtiles * P.ntiles;
    const int b = pid / per;
    pid -= b * per;
    P.A = (const char*)P.A + b * P.a_bs * 2;
    P.W = (const char*)P.W + b * P.w_bs * 2;
    P.C = (char*)P.C + b * P.c_bs_bytes;
    if (P.bias) P.bias += (long long)b * P.bias_bs;
    if (P.pscale) P.pscale += (long long)b * P.ps_ntiles * 512;
    P.batch_row0 = b * P.M;
  }
  if (const int gn = args.order & 0xff) {
    // weight-resident order: panels of gn column tiles, walked down the rows with the columns fastest -- an XCD's 32 concurrent
    // workgroups cover (32 / gn) row tiles x gn column tiles, the gn weight tiles stay in its L2 while the activation rows stream
    const int per_panel = gn * P.mtiles;
    const int panel = pid / per_panel;
    const int first_n = panel * gn;
    const int gsz = min(gn, P.ntiles - first_n);
    const int in_panel = pid - panel * per_panel;
    n0 = (first_n + in_panel % gsz) * TN;
    m0 = (in_panel / gsz) * TM;
    return P;
  }
  constexpr int GM = 8;
  const int per_panel = GM * P.ntiles;
  const int panel = pid / per_panel;
  const int first_m = panel * GM;
  const int gsz = min(GM, P.mtiles - first_m);
  const int in_panel = pid - panel * per_panel;
  m0 = (first_m + in_panel % gsz) * TM;
  n0 = (in_panel / gsz) * TN;
  return P;
}

// lane owns C[m][n .. n+3]: m = column (lane & 15) of tile j, n = 4 * (lane >> 4) + reg of tile i
// PRE: bias and residual are already inside the accumulators (accumulators_from_residual below)
// Every read (bias, residual, the GELU backward's pre-activation) is issued up front from clamped, always valid addresses and
// none sits under a branch: a conditional around a fragment's load makes the compiler close each fragment with vmcnt(0), i.e.
// one memory round trip per fragment in series (4-16 of them per wave on the layer chain's tiles).
template <typename T, int FN, int FM, int EPI, bool PRE = false>
__device__ __forceinline__ void epilogue(const GemmProb& P, f32x4 (&acc)[FN][FM], int n_base, int m_base, int lane) {
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  const int M = P.M;
  constexpr bool RES = EPI == EPI_RES_F32 && !PRE;
  int nc[FN];
  bool cok[FN];
  f32x4 bv[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n_base + i * 16 + ln;
    cok[i] = !P.n_mask || n < P.N;
    nc[i] = cok[i] ? n : 0;
    bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!PRE && P.bias) bv[i] = *reinterpret_cast<const f32x4*>(P.bias + nc[i]);
  }
  long long coff[FM], roff[FM];
  bool rok[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m_base + j * 16 + lm;
    rok[j] = m < M;
    const int mc = min(m, M - 1);
    coff[j] = EPI == EPI_KV ? 0 : view_off(P.c, mc);
    roff[j] = RES ? view_off(P.r, mc) : 0;
  }
  f32x4 rv[RES ? FN : 1][RES ? FM : 1];
  typename Vec4<T>::type uv[EPI == EPI_GELU_BWD ? FN : 1][EPI == EPI_GELU_BWD ? FM : 1];
  if constexpr (RES) {
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int i = 0; i < FN; ++i) rv[i][j] = *reinterpret_cast<const f32x4*>(P.R + roff[j] + nc[i]);
  }
  if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int i = 0; i < FN; ++i) uv[i][j] = *reinterpret_cast<const typename Vec4<T>::type*>((const T*)P.aux + coff[j] + nc[i]);
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    int item = 0, tok = 0;
    if constexpr (EPI == EPI_KV) {
      const int m = min(m_base + j * 16 + lm, M - 1);
      item = m / P.kv_tokens;
      tok = m - item * P.kv_tokens;
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = nc[i];
      const bool live = rok[j] && cok[i];
      f32x4 v = acc[i][j] + bv[i];
      if constexpr (EPI == EPI_GELU_OP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if constexpr (EPI == EPI_GELU_BOTH) {     // keep the pre-activation for the backward, return the activation
        typename Vec4<T>::type u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = from_f32<T>(v[e]); v[e] = gelu_erf((float)u[e]); }
        if (live) *reinterpret_cast<typename Vec4<T>::type*>((T*)P.aux + coff[j] + n) = u;
      }
      if constexpr (EPI == EPI_GELU_BWD) {      // d(pre-activation) = d(activation) * gelu'(pre-activation)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad((float)uv[i][j][e]);
      }
      if constexpr (RES) v += rv[i][j];
      if (!live) continue;
      if constexpr (EPI == EPI_RES_F32 || EPI == EPI_F32) {
        *reinterpret_cast<f32x4*>((float*)P.C + coff[j] + n) = v;
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
          *reinterpret_cast<typename Vec4<T>::type*>((T*)P.C + coff[j] + n) = o;
        }
      }
    }
  }
}

// The same epilogue with every read next to its use: for the loader-wave kernels, which sit at the 168-register cap of three waves per
// SIMD and have no room for the up-front loads (their 176 x 384 / 256 x 256 direct epilogues spilled 100-400 bytes per lane with them).
template <typename T, int FN, int FM, int EPI, bool PRE = false>
__device__ __forceinline__ void epilogue_lean(const GemmProb& P, f32x4 (&acc)[FN][FM], int n_base, int m_base, int lane) {
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  const int M = P.M;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m_base + j * 16 + lm;
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
      const int n = n_base + i * 16 + ln;
      if (P.n_mask && n >= P.N) continue;   // masked tail of a ragged last column tile
      f32x4 v = acc[i][j];
      if (!PRE && P.bias) v += *reinterpret_cast<const f32x4*>(P.bias + n);
      if constexpr (EPI == EPI_GELU_OP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if constexpr (EPI == EPI_GELU_BOTH) {     // keep the pre-activation for the backward, return the activation
        typename Vec4<T>::type u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = from_f32<T>(v[e]); v[e] = gelu_erf((float)u[e]); }
        *reinterpret_cast<typename Vec4<T>::type*>((T*)P.aux + coff + n) = u;
      }
      if constexpr (EPI == EPI_GELU_BWD) {      // d(pre-activation) = d(activation) * gelu'(pre-activation)
        const typename Vec4<T>::type u = *reinterpret_cast<const typename Vec4<T>::type*>((const T*)P.aux + coff + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad((float)u[e]);
      }
      if constexpr (EPI == EPI_RES_F32 && !PRE) v += *reinterpret_cast<const f32x4*>(P.R + roff + n);
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

// EPI_RES_F32 on the big tiles: start the accumulators at bias + residual instead of zero.  The fp32 residual read (half of
// that epilogue's traffic, 64-byte pieces of 16 rows per instruction) then happens before the K loop, hidden behind the wait
// for the first operand tiles, and the epilogue only stores.
template <int FN, int FM>
__device__ __forceinline__ void accumulators_from_residual(const GemmProb& P, f32x4 (&acc)[FN][FM], int n_base, int m_base, int lane) {
  // No load sits under a branch: with a conditional around each fragment the compiler ends every block in vmcnt(0) and the 32
  // residual loads of a 128 x 64 wave tile become 32 serial round trips (20-30 % of a K = 1408 tile).  Masked columns read a
  // clamped (valid) address and are zeroed afterwards.
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  int nc[FN];
  bool ok[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n_base + i * 16 + ln;
    ok[i] = !P.n_mask || n < P.N;
    nc[i] = ok[i] ? n : 0;
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = min(m_base + j * 16 + lm, P.M - 1);   // rows past M are computed on a clamped row and never stored
    const float* rrow = P.R + view_off(P.r, m);
#pragma unroll
    for (int i = 0; i < FN; ++i) acc[i][j] = *reinterpret_cast<const f32x4*>(rrow + nc[i]);
  }
  f32x4 bv[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.bias) bv[i] = *reinterpret_cast<const f32x4*>(P.bias + nc[i]);
  }
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = ok[i] ? acc[i][j] + bv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
}

// EPI_SOFTPART (scores of the folded cross-attention): first half of a softmax split over column tiles.  Needs a wave
// that holds whole tile rows (one column of compute waves): row m = lane & 15 of fragment j has its TN columns in the
// four lanes with that lane & 15, so the tile maximum and tile sum of a row are two lane swaps away.
template <typename T, int FN, int FM>
__device__ __forceinline__ void epilogue_softpart(const GemmProb& P, f32x4 (&acc)[FN][FM], int n0, int m_base, int tile, int batch_row0,
                                                  int lane) {
  // VALU-bound (67.6 k exponentials per 176 x 384 tile, 13 k cycles per SIMD): the scale rides the exponent's fma, and the
  // column-limit compares exist only in the last, ragged tile (wave-uniform branch)
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  const bool full = n0 + FN * 16 <= P.N;
  const float alpha = P.alpha;   // > 0
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m_base + j * 16 + lm;
    float mx = -3.0e38f;
    if (full) {
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, acc[i][j][e]);
    } else {
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + i * 16 + ln + e < P.N) mx = fmaxf(mx, acc[i][j][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    mx *= alpha;                     // max(alpha s) = alpha max(s), rounding included
    float l = 0.f;
    const bool live = m < P.M;
    const long long coff = live ? view_off(P.c, m) : 0;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + i * 16 + ln;
      typename Vec4<T>::type o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[i][j][e], alpha, -mx));
        if (!full && n + e >= P.N) p = 0.f;
        o[e] = from_f32<T>(p);
        l += (float)o[e];          // the sum of what the next GEMM will actually read
      }
      if (live) *reinterpret_cast<typename Vec4<T>::type*>((T*)P.C + coff + n) = o;
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (live && ln == 0) {
      const long long si = ((long long)batch_row0 + m) * P.ntiles + tile;
      P.stat_m[si] = mx;
      P.stat_l[si] = l;
    }
  }
}

// EPI_RES_F32_STAT: statistics of the FN * 16 = 128 columns a wave holds of each of its rows (row m = lane & 15 of fragment j: 32 values in
// the lane, the rest in the three lanes with the same lane & 15): group mean and sum of squared deviations FROM THAT MEAN -- two passes over
// registers, so nothing cancels however far the row's mean is from zero; groups are merged by launch_ln_group_stats (Chan et al.).
template <int FN, int FM>
__device__ __forceinline__ void group_stats(const GemmProb& P, f32x4 (&acc)[FN][FM], int n_base, int m_base, int lane) {
  static_assert(FN == 8, "a wave's tile must be 128 columns wide");
  const int lm = lane & 15;
  const int groups = P.N >> 7, g = n_base >> 7;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 128.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = acc[i][j][e] - mean; q = __builtin_fmaf(d, d, q); }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const int m = m_base + j * 16 + lm;
    if (lane < 16 && m < P.M) {
      float2* dst = reinterpret_cast<float2*>(P.ln_y32) + ((long long)m * groups + g);
      *dst = float2{mean, q};
    }
  }
}

// 16-bit outputs (EPI_OP, EPI_GELU_OP, EPI_KV) leave through LDS: a lane's natural store is 8 bytes
// of one output row, 16 rows per wave instruction -- 32-byte fragments of 128-byte lines; measured on
// the K/V projection that direct epilogue cost 23 k of the 105 k cycles of a 256x256 tile (store-issue
// bound, cdna_hip_programming.md T21).  Here the tile is first written to LDS as TN/64 column blocks of
// [TM rows][64 cols] (chunk index XOR (row & 7) against ds_write conflicts), then every thread moves
// 16-byte chunks out in LDS order: one wave instruction = 8 rows x 128 B, whole lines; for the
// head-major K/V cache that is 1 KiB contiguous.  Must be entered after a workgroup barrier (the K
// loop's LDS reads are over); smem needs TN * TM * 2 bytes.
// BIAS_FIRST: all FN bias pieces in one round trip before the staging loop (a load per fragment next to its use is one round trip
// per fragment); only on the small tiles: with 128 accumulators live the extra 4 FN registers spill (650-800 bytes per lane at the loader-wave
// kernels' 168-register budget, 300-350 in the eight-phase kernel).
// BIAS_LDS (the eight-phase tiles: 128 accumulators per lane, no room for FN bias pieces): the tile's TN bias values (and the folded LayerNorm's
// column sums) are fetched by TN threads in ONE round trip into LDS behind the staged tile (smem needs TN * TM * 2 + 8 TN bytes) and read from
// there per fragment.  Next to their use -- one conditional load per fragment, closed by vmcnt(0) -- they were eight serial L2 round trips
// per tile (the ISA showed `s_cbranch_execz; global_load; s_waitcnt vmcnt(0)` eight times in a row).
template <typename T, int TN, int TM, int FN, int FM, int NT, int EPI, bool BIAS_FIRST = false, bool BIAS_LDS = false>
__device__ __forceinline__ void epilogue_lds16(const GemmProb& P, f32x4 (&acc)[FN][FM], char* smem, int n0, int m0, int wn0,
                                               int wm0, int tid, bool stages = true) {   // stages == false (wave-uniform): this wave owns no accumulators, it only helps copying out
  const int lane = tid & 63;
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  static_assert(!(BIAS_FIRST && BIAS_LDS), "one way to fetch the bias");
  float* sbias = reinterpret_cast<float*>(smem + TN * TM * 2);   // [TN] bias, then [TN] column sums (BIAS_LDS)
  if constexpr (BIAS_LDS) {
    static_assert(TN <= NT, "one value per thread");
    if (tid < TN) {
      const int n = min(n0 + tid, P.N - 1);
      float b = P.bias ? P.bias[n] : 0.f;
      if (n0 + tid >= P.N) b = 0.f;
      sbias[tid] = b;
      if constexpr (EPI == EPI_LNF_OP || EPI == EPI_LNF_GELU_OP) sbias[TN + tid] = P.ln_gain[n];
    }
  }
  f32x4 bias4[BIAS_FIRST ? FN : 1];
  if constexpr (BIAS_FIRST) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn0 + i * 16 + ln;
      bias4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (P.bias) {
        bias4[i] = *reinterpret_cast<const f32x4*>(P.bias + (n < P.N ? n : 0));
        if (n >= P.N) bias4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  constexpr bool LNF = EPI == EPI_LNF_OP || EPI == EPI_LNF_GELU_OP;   // LayerNorm folded into this GEMM: see kernels.h
  float mu[LNF ? FM : 1], rstd[LNF ? FM : 1];
  if constexpr (LNF) {
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const float2 st = reinterpret_cast<const float2*>(P.ln_y32)[min(m0 + wm0 + j * 16 + lm, P.M - 1)];
      mu[j] = st.x; rstd[j] = st.y;
    }
  }
  if constexpr (BIAS_LDS) __syncthreads();
  if (stages)
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int nl = wn0 + i * 16 + ln;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS_FIRST) bv = bias4[i];
    else if constexpr (BIAS_LDS) bv = *reinterpret_cast<const f32x4*>(sbias + nl);
    else if (P.bias && n0 + nl < P.N) bv = *reinterpret_cast<const f32x4*>(P.bias + n0 + nl);
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LNF) {
      if constexpr (BIAS_LDS) cs = *reinterpret_cast<const f32x4*>(sbias + TN + nl);
      else cs = *reinterpret_cast<const f32x4*>(P.ln_gain + (n0 + nl < P.N ? n0 + nl : 0));
    }
    const int hq = nl >> 6, d = nl & 63;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int ml = wm0 + j * 16 + lm;
      f32x4 v;
      if constexpr (LNF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rstd[j], __builtin_fmaf(-mu[j], cs[e], acc[i][j][e]), bv[e]);
      } else {
        v = acc[i][j] + bv;
      }
      if constexpr (EPI == EPI_GELU_OP || EPI == EPI_LNF_GELU_OP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      typename Vec4<T>::type o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
      char* dst = smem + hq * (TM * 128) + ml * 128 + ((((d >> 3) ^ (ml & 7))) << 4) + ((d >> 2) & 1) * 8;
      *reinterpret_cast<typename Vec4<T>::type*>(dst) = o;
    }
  }
  __syncthreads();
  // Copy-out.  q = tid + k * NT walks the LDS image linearly; with TM * 8 = RPB * NT chunks per column
  // block a thread meets only RPB distinct rows (k % RPB) and one block per k / RPB, so the row's
  // (item, row-in-item) split -- one division for the tile, increments per row -- and the 64-bit
  // destination offsets are computed RPB times, not per chunk.
  constexpr int NCH = TN * TM / 8;  // 16-byte chunks in the tile
  static_assert(NCH % NT == 0 && (TM * 8) % NT == 0, "chunks must divide over the threads");
  constexpr int RPB = TM * 8 / NT;  // rows a thread touches per column block
  const int rpi = EPI == EPI_KV ? P.kv_tokens : P.c.rpi;
  const int item0 = m0 / rpi, r0 = m0 - item0 * rpi;
  const int c = (tid & 7);
  long long rowoff[RPB];
  bool live[RPB];
#pragma unroll
  for (int u = 0; u < RPB; ++u) {
    const int row = (tid >> 3) + u * (NT / 8);
    live[u] = m0 + row < P.M;
    int item = item0, r = r0 + (live[u] ? row : P.M - 1 - m0);   // dead rows: the address of the last live one (read, never stored)
    while (r >= rpi) { r -= rpi; ++item; }
    const int cc = c ^ (row & 7);
    if constexpr (EPI == EPI_KV) rowoff[u] = ((long long)item * P.kv_heads * P.kv_tokens + r) * 64 + cc * 8;
    else rowoff[u] = (long long)item * P.c.item_stride + (long long)r * P.c.ld + cc * 8;
  }
  // EPI_RES_OP: all residual chunks first (masked column blocks read block 0 of the tile), none under a branch -- a load per
  // (block, row) next to its add is one serial memory round trip each, 16 per thread on the 256 x 256 tile
  constexpr bool RESOP = EPI == EPI_RES_OP || EPI == EPI_RES_OP_STAT;
  typename Vec8<T>::type resid[RESOP ? TN / 64 : 1][RESOP ? RPB : 1];
  if constexpr (RESOP) {
#pragma unroll
    for (int hq = 0; hq < TN / 64; ++hq) {
      const int nb = n0 + hq * 64 < P.N ? n0 + hq * 64 : n0;
#pragma unroll
      for (int u = 0; u < RPB; ++u) resid[hq][u] = *reinterpret_cast<const typename Vec8<T>::type*>((const T*)P.aux + nb + rowoff[u]);
    }
  }
#pragma unroll
  for (int hq = 0; hq < TN / 64; ++hq) {
    const int nb = n0 + hq * 64;
    if (nb >= P.N) continue;   // n_mask: masked tail of a ragged last column tile (N % 64 == 0)
    long long blk;
    if constexpr (EPI == EPI_KV) {
      const int hidden = P.kv_heads * 64;
      const int sel = nb / hidden, head = (nb - sel * hidden) >> 6;
      blk = ((long long)sel * P.kv_items * P.kv_heads + head) * P.kv_tokens * 64;
    } else {
      blk = nb;
    }
#pragma unroll
    for (int u = 0; u < RPB; ++u) {
      if (!live[u]) continue;
      const int q = tid + (hq * RPB + u) * NT;
      typename Vec8<T>::type val = *reinterpret_cast<const typename Vec8<T>::type*>(smem + (size_t)q * 16);
      if constexpr (RESOP) {   // + the residual in the operand dtype, whole 16-byte chunks (the reference's fp16 add)
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = from_f32<T>((float)val[e] + (float)resid[hq][u][e]);
      }
      *reinterpret_cast<typename Vec8<T>::type*>((T*)P.C + blk + rowoff[u]) = val;
      if constexpr (EPI == EPI_RES_OP_STAT) {
        // the 64 columns of this row and block sit in the eight lanes tid & ~7 .. + 7 (all live or all dead together): mean and squared deviations
        // from it of the values AS STORED, two passes so that nothing cancels
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += (float)val[e];
        sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
        const float mean = sm * (1.0f / 64.0f);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)val[e] - mean; sq = __builtin_fmaf(d, d, sq); }
        sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
        if ((tid & 7) == 0) {
          const int m = m0 + (tid >> 3) + u * (NT / 8);
          reinterpret_cast<float2*>(P.ln_y32)[(long long)m * (P.N >> 6) + (nb >> 6)] = float2{mean, sq};
        }
      }
    }
  }
}

// =================================================================================================
// two-buffer main loop (64-deep K tiles, 128-byte rows): the 64x64 / 128x128 tiles of the layer chain
// =================================================================================================
template <typename T, int TN, int TM, int WGN, int WGM, int EPI>
__global__ void __launch_bounds__(WGN* WGM * 64) gemm_kernel(const GemmArgs args) {
  constexpr int BK = 64, ROWB = BK * 2;
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
  int n0, m0;
  const GemmProb& P = pick_tile<TN, TM>(args, n0, m0);
  const int K = P.K, M = P.M;

  // per-lane source pointers: physical chunk q & 7 of row q >> 3 holds source chunk (q & 7) ^ ((row >> 1) & 7)
  const char* srcW[IW];
  const char* srcX[IX];
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
    srcW[i] = (const char*)P.W + ((long long)min(n0 + row, P.N - 1) * K + c * 8) * 2;
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
  // fragment read offsets: row (lane & 15), logical chunk 4 * ks + (lane >> 4), swizzled
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
    __syncthreads();  // tile kt has landed for every wave; everyone is done reading the other buffer
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
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV || EPI == EPI_RES_OP) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, NT, EPI, (FN * FM <= 16)>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    if constexpr (FN * FM > 16) epilogue_lean<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);   // 256 x 256 on 8 waves (A/B variant only): no room for the up-front loads
    else epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
}

// =================================================================================================
// warp-specialised 256x256x64 main loop: 8 compute waves + 4 loader waves (768 threads)
// =================================================================================================
// Stamps on the two-buffer loop (K/V projection, 22 K tiles): per K tile a wave spends ~750 cycles
// issuing its 8 LDS-DMA instructions (the CU's one texture-address path takes all 64 KB serially) and
// ~650 at the barrier, against ~1900 in reads + MFMAs -- and the two waves of a SIMD do it in lockstep,
// so the matrix pipe idles 40 % of the loop.  Here the DMA issue moves to 4 dedicated loader waves
// (one per SIMD): the 8 compute waves (2 x 4, 128 x 64 outputs each) only read fragments and issue
// MFMAs between barriers; a loader issues tile kt+1 right after barrier kt and parks on vmcnt(0) until
// barrier kt+1, off the compute waves' critical path.  Three waves per SIMD cap the kernel at 168
// VGPRs: fragments are read just in time (all four B fragments, A one at a time).
// TN x TM = 256 x 256 (K/V projection) or 128 x 384 (the folded cross-attention's batched GEMMs: all 384
// (head, query) rows of an item against a 128-row slab of the other operand, so that operand streams once).
// 176 x 384 with the 8 compute waves in one column (WGM = 8): P . enc of the folded path -- N = 1408 = 8 x 176, so
// 32 items give exactly 256 workgroups, one per CU, and each streams its 176-row slab of enc^T exactly once.
template <typename T, int EPI, bool NODMA = false, int TN = 256, int TM = 256, int WGM = 4, bool NTW = false, bool WKM = false, bool PSC = false>  // NODMA: diagnostic only (wrong results); NTW: non-temporal loads of the weight-side slab; WKM: K-major W (GemmProb::w_ld); PSC: GemmProb::pscale
__global__ void __launch_bounds__(768) gemm_ws_kernel(const GemmArgs args) {
  constexpr int BK = 64, ROWB = BK * 2;
  constexpr int WGN = 8 / WGM, WTN = TN / WGN, WTM = TM / WGM, FN = WTN / 16, FM = WTM / 16;
  constexpr int BUF = (TN + TM) * ROWB;
  constexpr int PS_TILE = 176, GT_OFF = 2 * BUF, GT_SLICE = 2048;   // PSC: ring of four factor slices (one per 176-column tile of P~) behind the K buffers
  static_assert(!PSC || (TM <= 512 && PS_TILE % 8 == 0), "factor slices hold 512 rows; an 8-k fragment piece never straddles a tile");
  constexpr int NCHUNK = (TN + TM) * 8, NLD = (NCHUNK + 255) / 256;  // 16-byte chunks of a K tile (W rows, then A rows) / loader lanes
  constexpr bool STAGED = (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV || EPI == EPI_RES_OP) && TN % 64 == 0;   // LDS-staged 16-bit epilogue
  static_assert(WTN % 16 == 0 && WTM % 16 == 0 && TN % 16 == 0 && NCHUNK % 64 == 0, "tile must split over the waves; whole waves per DMA piece");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n0, m0;
  const GemmProb& P = pick_tile<TN, TM>(args, n0, m0);
  const int K = P.K, M = P.M;
  const int nk = K / BK;

  if (wave >= 8) {
    // ------------------------------- loader waves -------------------------------
    const int lt = tid - 512;  // 0..255
    const char* src[NLD];
    int wrow[NLD];             // WKM: K row of a weight-side chunk inside the tile
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = min(lt + i * 256, NCHUNK - 1);   // chunk of the K tile: rows [0, TN) are W, [TN, TN + TM) are A (TN % 16 == 0)
      const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
      wrow[i] = 0;
      if (WKM && q < TN * 8) {
        // K-major weights: the tile is [64 k][TN n], TN / 8 chunks per k row, no swizzle (fragments come from transposed reads)
        wrow[i] = q / (TN / 8);
        src[i] = (const char*)P.W + ((long long)n0 + (q - wrow[i] * (TN / 8)) * 8) * 2;
      } else if (row < TN) {
        const long long ldw = P.w_kwrap > 0 ? (long long)P.w_kwrap * BK : K;   // w_kwrap: W rows are only w_kwrap K steps long (K counts both passes)
        src[i] = (const char*)P.W + ((long long)min(n0 + row, P.N - 1) * ldw + c * 8) * 2;
      } else {
        const int m = min(m0 + row - TN, M - 1);
        src[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
      }
    }
    const int wq0 = (wave - 8) * 64;
    // GemmProb::w_kwrap: the weight-side operand is only w_kwrap K steps long and is walked again from its start (the hi | lo halves of a
    // split-precision activation row against the same weights: C = [A_hi | A_lo] . [W | W]^T without a second copy of W)
    const int kwrap = P.w_kwrap > 0 ? P.w_kwrap : 0x7fffffff;
    auto stage = [&](int buf, int kt) {
      char* base = smem + buf * BUF;
      const long long koff = (long long)kt * ROWB;
      const long long koffw = (long long)(kt >= kwrap ? kt - kwrap : kt) * ROWB;
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (i * 256 + wq0 < NCHUNK) {   // wave-uniform: whole 64-chunk pieces (8 rows; TN % 8 == 0)
          if (WKM && i * 256 + wq0 < TN * 8) {
            const long long krow = min(kt * BK + wrow[i], P.k_rows - 1);
            glds16_nt(src[i] + krow * P.w_ld * 2, base + (wq0 + i * 256) * 16);
          } else if ((i * 256 + wq0) / 8 < TN) {   // rows [0, TN): the weight side
            if (NTW) glds16_nt(src[i] + koffw, base + (wq0 + i * 256) * 16);   // read by this workgroup only
            else glds16(src[i] + koffw, base + (wq0 + i * 256) * 16);
          } else {
            glds16(src[i] + koff, base + (wq0 + i * 256) * 16);
          }
        }
    };
    // PSC: loader wave 8 also keeps the factor slices of the P~ tiles that K step kt + 1 touches in LDS (slot t & 3; a slot's previous
    // tenant, tile t - 4, was last read 8 K steps ago)
    int gt_next = 0;
    auto stage_factors = [&](int kt) {
      if constexpr (PSC) {
        if (wave != 8) return;
        const int t_hi = (kt * BK + BK - 1) / PS_TILE;
        for (; gt_next <= t_hi; ++gt_next) {   // past the last tile: its factors again (finite; the P~ columns there are zero)
          const char* sg = (const char*)(P.pscale + (long long)min(gt_next, P.ps_ntiles - 1) * 512) + lane * 16;
          char* dg = smem + GT_OFF + (gt_next & 3) * GT_SLICE;
          glds16(sg, dg);
          glds16(sg + 1024, dg + 1024);
        }
      }
    };
    stage(0, 0);
    stage_factors(0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 1 < nk && !(NODMA && kt >= 1)) { stage((kt + 1) & 1, kt + 1); stage_factors(kt + 1); }
    }
    if constexpr (STAGED) {
      __syncthreads();  // K loop reads over
      __syncthreads();  // tile staged in LDS by the compute waves
    }
    return;
  }

  // --------------------------------- compute waves ---------------------------------
  const int wn0 = (wave / WGM) * WTN;
  const int wm0 = (wave % WGM) * WTM;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int r = lane & 15;
    const int c = (4 * ks + (lane >> 4)) ^ ((r >> 1) & 7);
    foff[ks] = r * ROWB + c * 16;
  }
  // WKM: per-lane address inside the [64 k][TN n] weight tile for the transposed reads (row 8 g + (j >> 2), 4 columns from 4 (j & 3))
  const unsigned wtr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem +
                       (8 * (lane >> 4) + ((lane & 15) >> 2)) * (TN * 2) + (lane & 3) * 8;
  static_assert(!WKM || WGN == 1, "K-major weights: one column of compute waves");
  f32x4 acc[FN][FM];
  if constexpr (EPI == EPI_RES_F32) {
    accumulators_from_residual<FN, FM>(P, acc, n0 + wn0, m0 + wm0, lane);
  } else {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* wb = smem + (kt & 1) * BUF + wn0 * ROWB;
    const char* xb = smem + (kt & 1) * BUF + TN * ROWB + wm0 * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Vec8<T>::type b[FM];
#pragma unroll
      for (int j = 0; j < FM; ++j) b[j] = lds_read8<T>(xb + j * 16 * ROWB + foff[ks]);
      if constexpr (PSC) {
        // this lane's 8 k of the fragment lie inside one 176-column tile of P~: one factor per (row, tile), the rescale pass's arithmetic
        const int t = (kt * BK + ks * 32 + 8 * (lane >> 4)) / PS_TILE;
        const float* gt = reinterpret_cast<const float*>(smem + GT_OFF + (t & 3) * GT_SLICE) + wm0 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const float gf = gt[j * 16];
          if constexpr (sizeof(T) == 2 && __is_same(T, f16)) {
            // packed f16 multiplies (4 per fragment) with the factor rounded to f16: the fp32 form of the rescale pass costs 28 VALU
            // instructions per fragment and made this load-bound kernel compute-bound (0.236 -> 0.288 ms)
            b[j] = b[j] * (f16)gf;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) b[j][e] = from_f32<T>((float)b[j][e] * gf);
          }
        }
      }
      if constexpr (WKM) {
        // fragment of weight tile i: lane (column n = 16 i + (lane & 15)) needs k = 32 ks + 8 (lane >> 4) .. + 7 = two
        // transposed 4 x 16 blocks of the [k][n] tile; the next fragment's reads fly while this one's MFMAs issue
        constexpr int WP = TN * 2;
        const unsigned tb = wtr + (kt & 1) * BUF + ks * 32 * WP;
        i16x4 f[2][2];   // ping-pong fragment registers: indexed by constants after unrolling, so no moves of in-flight data
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(f[0][0]), "=&v"(f[0][1]) : "v"(tb), "n"(4 * WP) : "memory");
        static_assert(FN * 32 + 4 * WP < 65536, "fragment offsets must fit the ds_read offset field");
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          if (i + 1 < FN) {
            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                         : "=&v"(f[(i + 1) & 1][0]), "=&v"(f[(i + 1) & 1][1]) : "v"(tb), "n"((i + 1) * 32), "n"((i + 1) * 32 + 4 * WP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[i & 1][0]), "+v"(f[i & 1][1])::"memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[i & 1][0]), "+v"(f[i & 1][1])::"memory");
          }
          const i16x4 c0 = f[i & 1][0], c1 = f[i & 1][1];
          i16x8 v;
          v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3]; v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
          const typename Vec8<T>::type a_cur = __builtin_bit_cast(typename Vec8<T>::type, v);
#pragma unroll
          for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<T>(a_cur, b[j], acc[i][j]);
        }
      } else {
      typename Vec8<T>::type a_cur = lds_read8<T>(wb + foff[ks]);
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        typename Vec8<T>::type a_nxt = a_cur;
        if (i + 1 < FN) a_nxt = lds_read8<T>(wb + (i + 1) * 16 * ROWB + foff[ks]);
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<T>(a_cur, b[j], acc[i][j]);
        a_cur = a_nxt;
      }
      }
    }
  }
  if constexpr (EPI == EPI_SOFTPART) {
    static_assert(EPI != EPI_SOFTPART || WGN == 1, "the softmax-partial epilogue needs whole tile rows in one wave");
    epilogue_softpart<T, FN, FM>(P, acc, n0, m0 + wm0, n0 / TN, P.batch_row0, lane);
  } else if constexpr (STAGED) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, 512, EPI>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue_lean<T, FN, FM, EPI, EPI == EPI_RES_F32>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
}

// =================================================================================================
// 256x256x64, eight phases per pair of K tiles: 8 waves, both SIMD partners ping-pong between LDS reads + DMA issue and MFMAs
// =================================================================================================
// The schedule of cdna_hip_programming.md's "256^2 8-phase template", rebuilt from its rules (the example source is not in this
// image).  8 waves = 2 (weight-row halves, wr) x 4 (activation-row quarters, wc), 128 x 64 outputs each; the two waves of a SIMD are
// one of each wr and run one barrier apart (wr = 1 passes one barrier before the loop, wr = 0 one after it), so while one issues its 16
// MFMAs of a phase the other reads the next fragments and issues its share of the DMA.
// A K tile is staged as four 16 KB half-tiles, each read by EVERY wave in exactly one phase: W0 / W1 = the first / second 64 weight
// rows of each wr, X0 / X1 = the first / second 32 activation rows of each wc.  Per K tile and wave:
//   phase 1: read X0 (4 x b128), W0 (8)   MFMA W0 x X0        phase 3: read W1 (8, over W0)   MFMA W1 x X1
//   phase 2: read X1 (4)                  MFMA W0 x X1        phase 4: --                     MFMA W1 x X0   (X0 kept in registers)
// One half-tile (2 DMA instructions per wave) is staged per phase, as soon as its slot is dead: with E / O the even / odd LDS buffer,
//   phase 1: W1(O, this pair's odd tile)   2: X0(E, +2)   3: W0(E, +2)   4: X1(E, +2)   5: W1(E, +2)   6: X0(O, +3)   7: W0(O, +3)   8: X1(O, +3)
// and a counted vmcnt(6) in phases 4 and 8 leaves the three youngest half-tiles in flight: at phase 4 those are X1 / W0 / X0 of E(+2),
// so all of O has landed (read in phases 5-7); at phase 8 X1 / W0 / X0 of O(+3), so all of E(+2) has (read in the next phases 1-3).
// Reads of a retired buffer start one phase after the wait (every wave's wait + a barrier in between); a slot is restaged two
// phases after its last read (X0: one phase after, its reads are retired by lgkmcnt(8) before phase 1's first barrier).
// K / 64 must be even and >= 2.  Epilogues: those of the loader-wave kernel.
// TAIL: the same schedule on a 128 (weight rows) x 512 (activation rows) tile, for the half-width last column tile of N = 256 k + 128
// (the ViT's N = 1408 GEMMs): both wave groups share the 128 weight rows and take 256 activation rows each, so W0 / W1 are 8 KB (one DMA
// instruction per wave) and X0 / X1 32 KB (four): 80 KB per K tile, all 160 KB of LDS, vmcnt(9) = 4 + 1 + 4 for the three youngest
// half-tiles.  Reads, MFMAs and barriers are unchanged.
template <typename T, int EPI, bool TAIL>
__device__ __forceinline__ void gemm_p8_tile(const GemmProb& P, int n0, int m0, unsigned long long* dbg = nullptr) {
  constexpr int TN = TAIL ? 128 : 256, TM = TAIL ? 512 : 256, BK = 64, ROWB = BK * 2, WTN = 128, WTM = 64, FN = 8, FM = 4;
  constexpr int WHALF = (TAIL ? 64 : 128) * ROWB, XHALF = (TAIL ? 256 : 128) * ROWB;   // half-tile sizes: W0, W1 | X0, X1
  constexpr int BUF = 2 * WHALF + 2 * XHALF;                                           // 64 KB (80 KB) per K tile
  constexpr int WP = WHALF / 8192, XP = XHALF / 8192;                                  // DMA instructions per wave and half-tile
  constexpr bool STAGED = EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV || EPI == EPI_RES_OP || EPI == EPI_RES_OP_STAT || EPI == EPI_LNF_OP || EPI == EPI_LNF_GELU_OP;
  constexpr bool RESF32 = EPI == EPI_RES_F32 || EPI == EPI_RES_F32_STAT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int K = P.K, M = P.M;
  const int nk = K / BK;

#ifdef MRA_GEMM_EXPERIMENTS
  // stamped timeline (gemm_bench p8stamp): per workgroup and wave 8 x u64 of wall clock (100 MHz) at the phase boundaries
  unsigned long long* stamp = dbg ? dbg + ((size_t)blockIdx.x * 8 + wave) * 8 : nullptr;
#define P8_STAMP(i) do { if (stamp && lane == 0) stamp[i] = wall_clock64(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif
  P8_STAMP(0);
  // DMA sources: piece c of a half-tile is this lane's chunk q = tid + 512 c: LDS row q >> 3, physical chunk q & 7
  const char* srcw[2][WP];
  const char* srcx[2][XP];
#pragma unroll
  for (int c = 0; c < WP; ++c) {
    const int q = tid + c * 512;
    const int r = q >> 3, ch = (q & 7) ^ ((r >> 1) & 7);
    // 256-row tile: rows 0-63 | 128-191 (W0), 64-127 | 192-255 (W1); tail: rows 0-63 (W0), 64-127 (W1)
    const int w0row = TAIL ? r : (r < 64 ? r : 64 + r);
    srcw[0][c] = (const char*)P.W + ((long long)min(n0 + w0row, P.N - 1) * K + ch * 8) * 2;
    srcw[1][c] = (const char*)P.W + ((long long)min(n0 + w0row + 64, P.N - 1) * K + ch * 8) * 2;
  }
#pragma unroll
  for (int c = 0; c < XP; ++c) {
    const int q = tid + c * 512;
    const int r = q >> 3, ch = (q & 7) ^ ((r >> 1) & 7);
    const int x0row = (r >> 5) * 64 + (r & 31);              // 32 of every wave's 64 rows (wave index r >> 5: wc, or 4 wr + wc in the tail)
    srcx[0][c] = (const char*)P.A + (view_off(P.a, min(m0 + x0row, M - 1)) + ch * 8) * 2;
    srcx[1][c] = (const char*)P.A + (view_off(P.a, min(m0 + x0row + 32, M - 1)) + ch * 8) * 2;
  }
  auto stage = [&](int buf, int h, int kt) {   // h: 0 W0, 1 W1, 2 X0, 3 X1
    const long long koff = (long long)kt * ROWB;
    if (h < 2) {
      char* base = smem + buf * BUF + h * WHALF + wave * 1024;
#pragma unroll
      for (int c = 0; c < WP; ++c) glds16(srcw[h][c] + koff, base + c * 8192);
    } else {
      char* base = smem + buf * BUF + 2 * WHALF + (h - 2) * XHALF + wave * 1024;
#pragma unroll
      for (int c = 0; c < XP; ++c) glds16(srcx[h - 2][c] + koff, base + c * 8192);
    }
  };
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int r = lane & 15;
    const int c = (4 * ks + (lane >> 4)) ^ ((r >> 1) & 7);
    foff[ks] = r * ROWB + c * 16;
  }
  const int wrow = TAIL ? 0 : wr * 64 * ROWB, xrow = (TAIL ? wave : wc) * 32 * ROWB;   // this wave's rows inside a W / X half-tile

  f32x4 acc[FN][FM];
  if constexpr (RESF32) {
    // issued BEFORE the prologue's DMA: older in the vmcnt order, so the prologue's vmcnt(6) covers them and their latency
    // overlaps the first half-tiles' (an ordinary load still pending inside the loop would make the compiler drain everything)
    accumulators_from_residual<FN, FM>(P, acc, n0 + (TAIL ? 0 : wr * WTN), m0 + (TAIL ? wave : wc) * WTM, lane);
  } else {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  using V8 = typename Vec8<T>::type;
  V8 wf[4][2], x0[2][2], x1[2][2];
  auto read_w = [&](const char* half) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[i][ks] = lds_read8<T>(half + wrow + i * 16 * ROWB + foff[ks]);
  };
  auto read_x = [&](V8 (&x)[2][2], const char* half) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) x[j][ks] = lds_read8<T>(half + xrow + j * 16 * ROWB + foff[ks]);
  };
  auto mma = [&](int i0, int j0, V8 (&x)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i0 + i][j0 + j] = mfma16<T>(wf[i][ks], x[j][ks], acc[i0 + i][j0 + j]);
    __builtin_amdgcn_s_setprio(0);
  };
#define P8_BAR()                         \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");       \
  } while (0)

  // prologue: all of tile 0 (E) and X0 / W0 / X1 of tile 1 (O); W1(O) follows in phase 1
  stage(0, 2, 0); stage(0, 0, 0); stage(0, 3, 0); stage(0, 1, 0);
  stage(1, 2, 1); stage(1, 0, 1); stage(1, 3, 1);
  P8_STAMP(1);   // addresses computed, prologue DMA issued
  if constexpr (TAIL) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  P8_BAR();
  if (wr == 1) P8_BAR();   // the stagger
  P8_STAMP(2);   // first K tile landed

  const int npair = nk >> 1;
  for (int it = 0; it < npair; ++it) {
    if (it == 1) P8_STAMP(3);   // one pair of K tiles done
    if (it == npair - 1) P8_STAMP(4);   // before the last pair
    const int kt = 2 * it;
    const bool more = it + 1 < npair;        // another pair follows: its half-tiles are staged in phases 2-8
    char* E = smem;
    char* O = smem + BUF;
    // ---- phase 1 ----
    read_x(x0, E + 2 * WHALF);
    __builtin_amdgcn_sched_barrier(0);
    read_w(E);
    stage(1, 1, kt + 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the X0 reads (issued first) are done: X0(E) may be restaged next phase
    P8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(0, 0, x0);
    P8_BAR();
    // ---- phase 2 ----
    read_x(x1, E + 2 * WHALF + XHALF);
    if (more) stage(0, 2, kt + 2);
    P8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(0, 2, x1);
    P8_BAR();
    // ---- phase 3 ----
    read_w(E + WHALF);
    if (more) stage(0, 0, kt + 2);
    P8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(4, 2, x1);
    P8_BAR();
    // ---- phase 4 ----
    if (more) {
      stage(0, 3, kt + 2);
      if constexpr (TAIL) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8_BAR();
    mma(4, 0, x0);
    P8_BAR();
    // ---- phase 5 ----
    read_x(x0, O + 2 * WHALF);
    __builtin_amdgcn_sched_barrier(0);
    read_w(O);
    if (more) stage(0, 1, kt + 2);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    P8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(0, 0, x0);
    P8_BAR();
    // ---- phase 6 ----
    read_x(x1, O + 2 * WHALF + XHALF);
    if (more) stage(1, 2, kt + 3);
    P8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(0, 2, x1);
    P8_BAR();
    // ---- phase 7 ----
    read_w(O + WHALF);
    if (more) stage(1, 0, kt + 3);
    P8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(4, 2, x1);
    P8_BAR();
    // ---- phase 8 ----
    if (more) {
      stage(1, 3, kt + 3);
      if constexpr (TAIL) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    P8_BAR();
    mma(4, 0, x0);
    P8_BAR();
  }
  P8_STAMP(5);   // K loop done
  if (wr == 0) P8_BAR();   // re-join the two wave groups
#undef P8_BAR
  const int wn0 = TAIL ? 0 : wr * WTN, wm0 = (TAIL ? wave : wc) * WTM;
  if constexpr (STAGED) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, 512, EPI, false, true>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else if constexpr (EPI == EPI_RES_F32_STAT) {
    // the fp32 rows as EPI_RES_F32 stores them, the statistics of this wave's 128 columns of its 64 rows, then the same values once more in the
    // operand dtype through LDS (whole 128-byte lines): what the LayerNorm launch behind this GEMM used to read back and write
    epilogue<T, FN, FM, EPI_RES_F32, true>(P, acc, n0 + wn0, m0 + wm0, lane);
    group_stats<FN, FM>(P, acc, n0 + wn0, m0 + wm0, lane);
    __syncthreads();
    GemmProb Q = P;
    Q.C = P.ln_y16; Q.c = P.ln_y16v; Q.bias = nullptr;
    epilogue_lds16<T, TN, TM, FN, FM, 512, EPI_OP>(Q, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue<T, FN, FM, EPI, EPI == EPI_RES_F32>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
  P8_STAMP(6);   // epilogue issued
#undef P8_STAMP
}

template <typename T, int EPI, bool TAIL = false>
__global__ void __launch_bounds__(512) gemm_p8_kernel(const GemmArgs args) {
  int n0, m0;
#ifdef MRA_GEMM_EXPERIMENTS
  if (args.dbg && threadIdx.x == 0) args.dbg[((size_t)blockIdx.x * 8) * 8 + 7] = wall_clock64();   // kernel entry, before the tile lookup
#endif
  const GemmProb P = pick_tile<(TAIL ? 128 : 256), (TAIL ? 512 : 256)>(args, n0, m0);
  gemm_p8_tile<T, EPI, TAIL>(P, n0, m0, args.dbg);
}

// GemmProb::persist: the same tiles in the same order (the XCD remap sees the virtual block index), one workgroup per CU.  Between two tiles
// one barrier: the staged epilogue's LDS image has been read (its global stores may still be in flight) before the next tile's DMA lands in it.
template <typename T, int EPI>
__global__ void __launch_bounds__(512) gemm_p8_persist_kernel(const GemmArgs args) {
  const int nwg = args.total_tiles;
  const int q = nwg >> 3, r = nwg & 7;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
    const int xcd = vb & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    int n0, m0;
    const GemmProb P = tile_of<256, 256>(args, id, n0, m0);
    gemm_p8_tile<T, EPI, false>(P, n0, m0);
    __syncthreads();
  }
}

// N = 256 k + 128 (the ViT's N = 1408) without a masked half tile AND without a second pass over the activations: one launch, eleven
// workgroups per pair of 256-row tiles -- 2 x k full 256 x 256 tiles, then the pair's last 128 columns as one 128 x 512 tile -- in
// that order after the XCD remap, so the tiles that share activation rows run together and share them in L2.  (As its own launch
// the tail re-streams every activation row from HBM: measured slower than the masked tile.)  One problem, no batch.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_p8_mixed_tile(const GemmArgs& args, int vb) {
  int id = vb;
  {
    const int nwg = args.total_tiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const GemmProb& P = args.p[0];
  const int kfull = P.N >> 8, per = 2 * kfull + 1;     // full column tiles per row tile; workgroups per pair of row tiles
  const int pair = id / per, w = id - pair * per;
  if (w == 2 * kfull) {
    gemm_p8_tile<T, EPI, true>(P, kfull * 256, pair * 512);
  } else {
    const int rt = 2 * pair + w / kfull;
    if (rt * 256 >= P.M) return;                         // odd number of row tiles: the last pair has one
    gemm_p8_tile<T, EPI, false>(P, (w % kfull) * 256, rt * 256);
  }
}
template <typename T, int EPI>
__global__ void __launch_bounds__(512) gemm_p8_mixed_kernel(const GemmArgs args) {
  gemm_p8_mixed_tile<T, EPI>(args, blockIdx.x);
}

// =================================================================================================
// ring: one workgroup per CU on an exact-fit tile, K tiles through a ring of STAGES LDS slots, two wave groups on alternating K tiles
// =================================================================================================
// At ~2 k activation rows a launch is ONE round of the chip: its time is the time of its slowest CU, and with a single workgroup on a
// CU nothing overlaps by occupancy.  Stamped timelines (tests/native/gemm_bench stamp; QKV 2048 x 768 -> 2304) of the forms tried:
//   two-buffer loop, 288 tiles of 128 x 128                                   17.6 us per launch
//   ring, all waves stage / read / multiply in lockstep (6 waves, 3 in flight) 16.0 us: per 64-deep step DMA issue ~400 + fragment reads
//                                                                              ~650 + MFMAs ~770 cycles in series
//   two groups by K parity, one group staging, L2 "touch" warm-up of the tile  19.1 us: touches 1.3-2.9 us (64 lines per wave instruction
//                                                                              serialise on the CU's address path; no gain behind them),
//                                                                              first data at 4.5 us, 0.54 us per interval, group reduction
//                                                                              1.2 us, direct 8-byte epilogue stores by four waves 3.4 us
// What is kept:
//   * tile shapes that give EXACTLY one workgroup per CU at the chain's shapes (QKV 2048 x 2304 = 16 x 16 tiles of 128 rows x 144 weight
//     rows; FFN-up 2 x 1024 x 3072 = 2 x 8 x 16 tiles of 128 x 192; the N = 768 projections 64 rows x 96 weight rows): half the operand
//     bytes per CU of 64 x 64 tiles;
//   * 2 x NWC waves = two groups that each own the WHOLE tile for every other K tile (wave w and wave w + NWC share a SIMD).  One barrier
//     per K tile opens an interval in which one group reads the tile's fragments from LDS into registers while its SIMD partners run
//     the MFMAs of the previous tile; every wave stages its share of the tile STAGES - 2 ahead (counted vmcnt: STAGES - 3 tiles stay in
//     flight across the barrier), the readers behind their reads, the multipliers behind their MFMAs;
//   * epilogue: group 1's partial sums go to group 0 through the ring's space; group 0 finishes the values (bias, GELU) and lays the tile
//     out row-major in LDS; ALL waves copy it out in 16-byte pieces of whole rows (fp32 outputs too; the fp32 residual was loaded into
//     the accumulators before the K loop).
// Slot of tile t: t % STAGES; it is re-staged in interval t + 2, after its readers passed the barrier that follows their MFMAs' operand wait.
template <int N>
__device__ __forceinline__ void ring_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int TN, int TM, int WGN, int WGM, int STAGES, int EPI>
__global__ void __launch_bounds__(2 * WGN* WGM * 64) gemm_ring_kernel(const GemmArgs args) {
  constexpr int BK = 64, ROWB = BK * 2;
  constexpr int NWC = WGN * WGM, NW = 2 * NWC, NT = NW * 64;
  constexpr int WTN = TN / WGN, WTM = TM / WGM, FN = WTN / 16, FM = WTM / 16;
  constexpr int ROWS = TN + TM, NPIECE = ROWS / 8, PPW = (NPIECE + NW - 1) / NW;   // 1 KiB pieces per K tile / per wave (upper bound)
  constexpr int BUF = ROWS * ROWB;
  constexpr bool RESID = EPI == EPI_RES_F32 || EPI == EPI_RES_LN;
  constexpr bool F32OUT = RESID || EPI == EPI_F32;
  constexpr int OSZ = F32OUT ? 4 : 2, PITCH = TN * OSZ + 16;                         // staged output tile: row-major, padded rows
  constexpr int RED_BYTES = NWC * FN * FM * 1024;
  static_assert(TN % (16 * WGN) == 0 && TM % (16 * WGM) == 0 && ROWS % 8 == 0 && TN % 8 == 0, "tile must split over the waves; whole pieces");
  static_assert(STAGES >= 4 && (STAGES - 3) * PPW <= 63, "vmcnt holds 6 bits");
  static_assert(RED_BYTES + TM * PITCH <= STAGES * BUF, "reduction + staged tile reuse the ring");
  static_assert(EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_RES_F32 || EPI == EPI_F32 || EPI == EPI_RES_LN, "epilogues of the ring kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave / NWC, gw = wave - group * NWC;
  const int wn0 = (gw / WGM) * WTN;
  const int wm0 = (gw % WGM) * WTM;
  int n0, m0;
  const GemmProb& P = pick_tile<TN, TM>(args, n0, m0);
  const int K = P.K, M = P.M;
  const int nk = K / BK;
#ifdef MRA_GEMM_EXPERIMENTS
  // stamped timeline (gemm_bench stamp): per workgroup and wave 8 x u64 of wall clock (100 MHz) / shader clock at the phase boundaries
  unsigned long long* stamp = args.dbg ? args.dbg + ((size_t)blockIdx.x * NW + wave) * 16 : nullptr;
#define RING_STAMP(i) do { if (stamp && lane == 0) { stamp[i] = wall_clock64(); stamp[8 + i] = clock64(); } } while (0)
#else
#define RING_STAMP(i) do { } while (0)
#endif
  RING_STAMP(0);

  // this wave's pieces of every K tile: p = wave + i * NW; piece p = LDS rows 8 p .. 8 p + 7 (rows [0, TN) weights, then activations)
  const char* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = min(wave + i * NW, NPIECE - 1);
    const int row = p * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    if (row < TN) {   // wave-uniform: TN % 8 == 0
      src[i] = (const char*)P.W + ((long long)min(n0 + row, P.N - 1) * K + c * 8) * 2;
    } else {
      const int m = min(m0 + row - TN, M - 1);  // rows past M are computed on a clamped row and never stored
      src[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
    }
  }
  const bool full = wave + (PPW - 1) * NW < NPIECE;   // this wave issues PPW pieces per K tile (else PPW - 1)
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * BUF + wave * 1024;
    const long long koff = (long long)kt * ROWB;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      if (i + 1 < PPW || full) glds16(src[i] + koff, base + i * (NW * 1024));
  };
  // the first STAGES - 2 tiles go out before anything else is computed
#pragma unroll
  for (int t = 0; t < STAGES - 2; ++t)
    if (t < nk) stage(t, t);
  RING_STAMP(1);
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int r = lane & 15;
    const int c = (4 * ks + (lane >> 4)) ^ ((r >> 1) & 7);
    foff[ks] = r * ROWB + c * 16;
  }
  f32x4 acc[FN][FM];
  if (RESID && group == 0) {
    // ordinary loads behind the prologue's DMA: the bias add below makes the compiler wait for them with vmcnt(0), which also covers the
    // prologue's tiles (one round trip for both); nothing of them stays in the queue the loop counts
    accumulators_from_residual<FN, FM>(P, acc, n0 + wn0, m0 + wm0, lane);
  } else {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  RING_STAMP(2);

  using V8 = typename Vec8<T>::type;
  V8 a[2][FN], b[2][FM];
  int rslot = 0, fslot = STAGES - 2;   // slots of tile kt and of tile kt + STAGES - 2
  // One interval: the barrier that publishes tile kt, then this group's role.  G (the group) and PAR (kt & 1) are compile-time at
  // each call site, so each group runs a straight-line body (a run-time role switch made the compiler merge the fragment registers of
  // both roles through copies: 256 VGPRs + spills inside the loop).
  auto interval = [&](auto G, auto PAR, int kt) {
    constexpr int g = decltype(G)::value, par = decltype(PAR)::value;
    // tile kt must have landed; tiles kt + 1 .. kt + STAGES - 3 (issued in the intervals since) may stay in flight
    if (kt < nk) {
      if (STAGES > 3 && kt + STAGES - 3 < nk) {
        if (full) ring_wait_vmcnt<(STAGES - 3) * PPW>(); else ring_wait_vmcnt<(STAGES - 3) * (PPW - 1)>();
      } else {
        ring_wait_vmcnt<0>();
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt == 0) RING_STAMP(3);
    if (kt == 4) RING_STAMP(4);
    if constexpr (par == g) {
      if (kt < nk) {   // this group's turn: the fragments of tile kt into registers (waited for by the MFMAs of the next interval)
        const char* wb = smem + rslot * BUF + wn0 * ROWB;
        const char* xb = smem + rslot * BUF + TN * ROWB + wm0 * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < FN; ++i) a[ks][i] = lds_read8<T>(wb + i * 16 * ROWB + foff[ks]);
#pragma unroll
          for (int j = 0; j < FM; ++j) b[ks][j] = lds_read8<T>(xb + j * 16 * ROWB + foff[ks]);
        }
      }
    } else {
      if (kt >= 1) {   // the MFMAs of tile kt - 1 (read in the previous interval)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<T>(a[ks][i], b[ks][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    if (kt + STAGES - 2 < nk) stage(fslot, kt + STAGES - 2);   // behind the reads / the MFMAs: the slot's last readers passed this barrier
    rslot = rslot + 1 == STAGES ? 0 : rslot + 1;
    fslot = fslot + 1 == STAGES ? 0 : fslot + 1;
  };
  auto run = [&](auto G) {
#pragma clang loop unroll(disable)
    for (int kt = 0; kt <= nk; kt += 2) {
      interval(G, std::integral_constant<int, 0>{}, kt);
      if (kt + 1 <= nk) interval(G, std::integral_constant<int, 1>{}, kt + 1);
    }
  };
  if (group == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
  RING_STAMP(5);

  // ---- epilogue: the groups exchange halves of their partial sums (fragment f = i * FM + j belongs to group f & 1), each finishes its
  //      half (bias, GELU, conversion) into a row-major tile in LDS, then every wave copies whole-row pieces out ----
  // the bias pieces ride the round trip of the exchange below (read next to their use they cost ~1 us of exposed latency: stamps r03g)
  f32x4 bvs[FN];
  {
    const int ln = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      bvs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!RESID && P.bias) bvs[i] = *reinterpret_cast<const f32x4*>(P.bias + n0 + wn0 + i * 16 + ln);   // RES_F32 / RES_LN: bias and residual are inside group 0's accumulators
    }
  }
  __syncthreads();   // the ring is dead
  {
    static_assert((FN * FM) % 2 == 0, "fragments split evenly over the two groups");
    constexpr int HALF = FN * FM / 2;
    // outgoing fragments of group g of wave pair gw: [g][gw][HALF][64 lanes] f32x4
    f32x4* out_red = reinterpret_cast<f32x4*>(smem) + ((size_t)(group * NWC + gw) * HALF) * 64 + lane;
    const f32x4* in_red = reinterpret_cast<const f32x4*>(smem) + ((size_t)((group ^ 1) * NWC + gw) * HALF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j)
        if (((i * FM + j) & 1) != group) out_red[((i * FM + j) >> 1) * 64] = acc[i][j];
    __syncthreads();
    char* tile = smem + RED_BYTES;
    const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int nl = wn0 + i * 16 + ln;
      const f32x4 bv = bvs[i];
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        if (((i * FM + j) & 1) != group) continue;
        f32x4 v = acc[i][j] + in_red[((i * FM + j) >> 1) * 64] + bv;
        if constexpr (EPI == EPI_GELU_OP) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        char* dst = tile + (wm0 + j * 16 + lm) * PITCH + nl * OSZ;
        if constexpr (F32OUT) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          typename Vec4<T>::type o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
          *reinterpret_cast<typename Vec4<T>::type*>(dst) = o;
        }
      }
    }
  }
  RING_STAMP(6);
  __syncthreads();
  {
    constexpr int CPR = TN * OSZ / 16;            // 16-byte pieces per output row
    constexpr int NCH = TM * CPR;
    const char* tile = smem + RED_BYTES;
    for (int q = tid; q < NCH; q += NT) {
      const int row = q / CPR, c = q - row * CPR;
      if (m0 + row >= M) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * PITCH + c * 16);
      char* out = (char*)P.C + (view_off(P.c, m0 + row) + n0) * OSZ + c * 16;
      if constexpr (EPI == EPI_RES_LN) {
        // read by ANOTHER workgroup, possibly on another XCD (whose L2 is not coherent with this one's): sc1 = written through
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(out), "v"(v) : "memory");
      } else {
        *reinterpret_cast<f32x4*>(out) = v;
      }
    }
  }
  if constexpr (EPI == EPI_RES_LN) {
    // LayerNorm by the last-arriving column tile of this row block (MI355X_MICROARCH.md, inter-workgroup visibility: every store sc1 and
    // drained by its wave, the workgroup's barrier, ONE agent-scope atomic add; the workgroup whose add came last -- told by the value
    // returned -- reads the bytes with sc1 loads after a barrier behind that add).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem);   // the reduction space is dead
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(P.ln_counter + m0 / TM, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = old == (unsigned)(P.ntiles - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (*flag) {
      // every wave normalises TM / NW rows; ALL their loads go out before the one wait (a round trip per row in series cost ~2 us each)
      constexpr int RPW = (TM + NW - 1) / NW;
      const int H = P.N;                       // 256 * NV columns, NV = 1 .. 4
      const int nv = H >> 8;
      f32x4 v[RPW][4];
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        const int m = min(m0 + wave + u * NW, M - 1);
        const float* xr = (const float*)P.C + view_off(P.c, m);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (i < nv) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u][i]) : "v"(xr + (i * 64 + lane) * 4) : "memory");
        }
      }
      f32x4 g[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = min(i, nv - 1) * 256 + lane * 4;
        g[i] = *reinterpret_cast<const f32x4*>(P.ln_gain + c);
        b[i] = *reinterpret_cast<const f32x4*>(P.ln_bias + c);
      }
#pragma unroll
      for (int u = 0; u < RPW; ++u)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[u][0]), "+v"(v[u][1]), "+v"(v[u][2]), "+v"(v[u][3])::"memory");
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        const int r = wave + u * NW;
        const int m = m0 + r;
        if (r >= TM || m >= M) continue;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nv) sum += (v[u][i][0] + v[u][i][1]) + (v[u][i][2] + v[u][i][3]);
        const float mean = wave_sum(sum) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[u][i][e] -= mean; q += v[u][i][e] * v[u][i][e]; }
          }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + P.ln_eps);
        float* o32 = P.ln_y32 ? P.ln_y32 + view_off(P.ln_y32v, m) : nullptr;
        T* o16 = P.ln_y16 ? (T*)P.ln_y16 + view_off(P.ln_y16v, m) : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            f32x4 y;
            typename Vec4<T>::type y16;
#pragma unroll
            for (int e = 0; e < 4; ++e) { y[e] = v[u][i][e] * rstd * g[i][e] + b[i][e]; y16[e] = from_f32<T>(y[e]); }
            if (o32) *reinterpret_cast<f32x4*>(o32 + c) = y;
            if (o16) *reinterpret_cast<typename Vec4<T>::type*>(o16 + c) = y16;
          }
      }
      if (tid == 0) P.ln_counter[m0 / TM] = 0u;   // ready for the next launch (visible at the launch boundary)
    }
  }
  RING_STAMP(7);
#undef RING_STAMP
}

// =================================================================================================
// k128: two 128-deep buffers (256-byte rows) for the small projections (M <= 2048 rows)
// =================================================================================================
// The 64x64 / 128x128 launches of the 12-layer chain are latency-bound per K step (wait -> barrier ->
// DMA issue -> fragment reads -> 8 MFMAs: ~1100 cycles for 128 cycles of MFMA work); halving the
// number of steps is worth more than anything inside a step.  256-byte rows: chunk index XOR (row & 15)
// makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots.
template <typename T, int TN, int TM, int WGN, int WGM, int EPI>
__global__ void __launch_bounds__(WGN* WGM * 64) gemm_k128_kernel(const GemmArgs args) {
  constexpr int BK = 128, ROWB = BK * 2, CPR = 16;
  constexpr int NT = WGN * WGM * 64;
  constexpr int WTN = TN / WGN, WTM = TM / WGM;
  constexpr int FN = WTN / 16, FM = WTM / 16;
  constexpr int IW = TN * CPR / NT, IX = TM * CPR / NT;
  static_assert(TN * CPR % NT == 0 && TM * CPR % NT == 0, "tile/threads mismatch");
  constexpr int BUF = (TN + TM) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn0 = (wave / WGM) * WTN;
  const int wm0 = (wave % WGM) * WTM;
  int n0, m0;
  const GemmProb& P = pick_tile<TN, TM>(args, n0, m0);
  const int K = P.K, M = P.M;
  const char* srcW[IW];
  const char* srcX[IX];
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 4, c = (q & 15) ^ (row & 15);
    srcW[i] = (const char*)P.W + ((long long)min(n0 + row, P.N - 1) * K + c * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 4, c = (q & 15) ^ (row & 15);
    const int m = min(m0 + row, M - 1);
    srcX[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
  }
  const int wave_q0 = wave * 64;
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * BUF;
    const long long koff = (long long)kt * ROWB;
#pragma unroll
    for (int i = 0; i < IW; ++i) glds16(srcW[i] + koff, base + (wave_q0 + i * NT) * 16);
#pragma unroll
    for (int i = 0; i < IX; ++i) glds16(srcX[i] + koff, base + TN * ROWB + (wave_q0 + i * NT) * 16);
  };
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int r = lane & 15;
    foff[ks] = r * ROWB + (((4 * ks + (lane >> 4)) ^ r) << 4);
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
    for (int ks = 0; ks < 4; ++ks) {
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
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV || EPI == EPI_RES_OP) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, NT, EPI, (FN * FM <= 16)>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
}

// launches per (kernel family, epilogue) since the library was loaded: read-only diagnostics (mra_debug_gemm_launches) so that a parity
// test can state WHICH main loop produced the numbers it checked
std::atomic<long long> g_launches[GEMM_FAMILIES][16];
inline int counted(int family, int epi, int rc) {
  if (rc == 0 && family >= 0 && family < GEMM_FAMILIES && epi >= 0 && epi < 16) g_launches[family][epi].fetch_add(1, std::memory_order_relaxed);
  return rc;
}

#ifdef MRA_GEMM_EXPERIMENTS
int g_force_cfg = -1;
int g_variant = 5;  // 5 (default): warp-specialised 256x256, two-buffer small tiles (128-deep for K >= 2048);
                    // 1: two-buffer loop everywhere; other values: gemm_experiments.inc (experiment builds only)
unsigned long long* g_dbg = nullptr;
int g_p8 = 1;       // the eight-phase kernel for the 256 x 256 tile when K / 64 is even (0: the loader-wave kernel, for A/B runs)
int g_order = 0;
#else
// the shipped library: no switches, the defaults are constants
constexpr int g_force_cfg = -1, g_variant = 5, g_p8 = 1, g_order = 0;
constexpr unsigned long long* g_dbg = nullptr;
#endif

// hipFuncSetAttribute once per kernel and device (it is not a stream operation: keep it out of the
// per-launch path and out of graph captures)
bool ensure_lds(const void* fn, size_t lds) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  if (done.count({fn, dev})) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
  done.insert({fn, dev});
  return true;
}

template <typename KFN>
int launch_k(KFN kfn, const GemmArgs& a, int threads, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024 && !ensure_lds((const void*)kfn, lds)) return -3;
  hipLaunchKernelGGL(kfn, dim3(a.total_tiles), dim3(threads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

#define MRA_EPI_SWITCH(KERNEL_EXPR)                         \
  switch (epi) {                                            \
    case EPI_OP: { constexpr int E = EPI_OP; return KERNEL_EXPR; }           \
    case EPI_GELU_OP: { constexpr int E = EPI_GELU_OP; return KERNEL_EXPR; } \
    case EPI_RES_F32: { constexpr int E = EPI_RES_F32; return KERNEL_EXPR; } \
    case EPI_F32: { constexpr int E = EPI_F32; return KERNEL_EXPR; }         \
    case EPI_KV: { constexpr int E = EPI_KV; return KERNEL_EXPR; }           \
    case EPI_GELU_BOTH: { constexpr int E = EPI_GELU_BOTH; return KERNEL_EXPR; } \
    case EPI_GELU_BWD: { constexpr int E = EPI_GELU_BWD; return KERNEL_EXPR; }   \
    case EPI_RES_OP: { constexpr int E = EPI_RES_OP; return KERNEL_EXPR; }       \
    default: return -2;                                     \
  }

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_v1(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 128;
  MRA_EPI_SWITCH((launch_k(gemm_kernel<T, TN, TM, WGN, WGM, E>, a, WGN * WGM * 64, lds, stream)))
}

template <typename T>
int launch_ws(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (256 + 256) * 128;
  MRA_EPI_SWITCH((launch_k(gemm_ws_kernel<T, E>, a, 768, lds, stream)))
}

template <typename T>
int launch_p8(const GemmArgs& a, int epi, hipStream_t stream, bool tail = false) {
  constexpr size_t lds = 2 * (256 + 256) * 128 + 8 * 256;   // two K-tile buffers; the staged epilogues keep the tile's bias values behind the 128 KB tile image
  if (tail) {   // 128 x 512 tile: 2 x 80 KB
    constexpr size_t ldst = 2 * (128 + 512) * 128;
    switch (epi) {
      case EPI_RES_OP: return launch_k(gemm_p8_kernel<T, EPI_RES_OP, true>, a, 512, ldst, stream);
      case EPI_RES_OP_STAT: return launch_k(gemm_p8_kernel<T, EPI_RES_OP_STAT, true>, a, 512, ldst, stream);
      case EPI_RES_F32: return launch_k(gemm_p8_kernel<T, EPI_RES_F32, true>, a, 512, ldst, stream);
      case EPI_RES_F32_STAT: return launch_k(gemm_p8_kernel<T, EPI_RES_F32_STAT, true>, a, 512, ldst, stream);
      case EPI_F32: return launch_k(gemm_p8_kernel<T, EPI_F32, true>, a, 512, ldst, stream);
      default: return -2;
    }
  }
  if (a.p[0].persist && a.ngroups == 1 && a.p[0].batch <= 1) {   // one workgroup per CU (the grid must be a multiple of the 8 XCDs for the remap to hold)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8 && a.total_tiles > cus) {
      const int grid = cus & ~7;
      auto go = [&](auto kfn) {
        if (!ensure_lds((const void*)kfn, lds)) return -3;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, stream, a);
        return hipGetLastError() == hipSuccess ? 0 : -4;
      };
      switch (epi) {
        case EPI_OP: return go(gemm_p8_persist_kernel<T, EPI_OP>);
        case EPI_GELU_OP: return go(gemm_p8_persist_kernel<T, EPI_GELU_OP>);
        case EPI_LNF_OP: return go(gemm_p8_persist_kernel<T, EPI_LNF_OP>);
        case EPI_LNF_GELU_OP: return go(gemm_p8_persist_kernel<T, EPI_LNF_GELU_OP>);
        case EPI_KV: return go(gemm_p8_persist_kernel<T, EPI_KV>);
        default: break;   // the direct epilogues: one workgroup per tile
      }
    }
  }
  switch (epi) {
    case EPI_OP: return launch_k(gemm_p8_kernel<T, EPI_OP>, a, 512, lds, stream);
    case EPI_GELU_OP: return launch_k(gemm_p8_kernel<T, EPI_GELU_OP>, a, 512, lds, stream);
    case EPI_KV: return launch_k(gemm_p8_kernel<T, EPI_KV>, a, 512, lds, stream);
    case EPI_RES_OP: return launch_k(gemm_p8_kernel<T, EPI_RES_OP>, a, 512, lds, stream);
    case EPI_RES_OP_STAT: return launch_k(gemm_p8_kernel<T, EPI_RES_OP_STAT>, a, 512, lds, stream);
    case EPI_RES_F32: return launch_k(gemm_p8_kernel<T, EPI_RES_F32>, a, 512, lds, stream);
    case EPI_RES_F32_STAT: return launch_k(gemm_p8_kernel<T, EPI_RES_F32_STAT>, a, 512, lds, stream);
    case EPI_LNF_OP: return launch_k(gemm_p8_kernel<T, EPI_LNF_OP>, a, 512, lds, stream);
    case EPI_LNF_GELU_OP: return launch_k(gemm_p8_kernel<T, EPI_LNF_GELU_OP>, a, 512, lds, stream);
    case EPI_F32: return launch_k(gemm_p8_kernel<T, EPI_F32>, a, 512, lds, stream);
    default: return -100;
  }
}

template <typename T>
int launch_ws_fold(const GemmArgs& a, int epi, hipStream_t stream) {   // 128 (weight rows) x 384 (activation rows)
  constexpr size_t lds = 2 * (128 + 384) * 128;
  switch (epi) {
    case EPI_OP: return launch_k(gemm_ws_kernel<T, EPI_OP, false, 128, 384>, a, 768, lds, stream);
    case EPI_F32: return launch_k(gemm_ws_kernel<T, EPI_F32, false, 128, 384>, a, 768, lds, stream);
    default: return -2;
  }
}

template <typename T>
int launch_ws_pv(const GemmArgs& a, int epi, hipStream_t stream) {   // 176 (weight rows) x 384 (activation rows), 1 x 8 waves
  constexpr size_t lds = 2 * (176 + 384) * 128;
  if (a.p[0].w_ld > 0) {   // K-major weights (P . enc straight from the encoder tokens)
    if (epi != EPI_OP) return -2;
    if (a.p[0].pscale) {
      if (a.p[0].M > 384 || a.p[0].ps_ntiles <= 0) return -1;   // one row tile: the factor slice is indexed by the row inside the tile
      return launch_k(gemm_ws_kernel<T, EPI_OP, false, 176, 384, 8, true, true, true>, a, 768, lds + 4 * 2048, stream);
    }
    return launch_k(gemm_ws_kernel<T, EPI_OP, false, 176, 384, 8, true, true>, a, 768, lds, stream);
  }
  // the slab (weight-side rows) is read by this workgroup only: non-temporal loads (7.40 -> 7.17 ms / step)
  if (epi == EPI_F32) return launch_k(gemm_ws_kernel<T, EPI_F32, false, 176, 384, 8, true>, a, 768, lds, stream);
  if (epi == EPI_SOFTPART) return launch_k(gemm_ws_kernel<T, EPI_SOFTPART, false, 176, 384, 8, true>, a, 768, lds, stream);
  if (epi == EPI_OP) return launch_k(gemm_ws_kernel<T, EPI_OP, false, 176, 384, 8, true>, a, 768, lds, stream);
  return -2;
}

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_k128(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 256;
  MRA_EPI_SWITCH((launch_k(gemm_k128_kernel<T, TN, TM, WGN, WGM, E>, a, WGN * WGM * 64, lds, stream)))
}

// the ring tiles (GemmProb::tile_cfg 9 / 10 / 11): 144 x 128, 192 x 128, 96 x 64 on 2 x 4 waves.  An epilogue whose staged tile does not fit
// beside the group reduction in the ring's LDS (fp32 outputs of the two large tiles) is refused.
template <int TN, int TM, int WGN, int WGM, int STAGES, int EPI>
constexpr bool ring_fits() {
  constexpr int osz = (EPI == EPI_RES_F32 || EPI == EPI_F32 || EPI == EPI_RES_LN) ? 4 : 2;
  return WGN * WGM * (TN / WGN / 16) * (TM / WGM / 16) * 1024 + TM * (TN * osz + 16) <= STAGES * (TN + TM) * 128;
}
template <typename T, int TN, int TM, int WGN, int WGM, int STAGES, int EPI>
int launch_ring_epi(const GemmArgs& a, hipStream_t stream) {
  if constexpr (ring_fits<TN, TM, WGN, WGM, STAGES, EPI>()) {
    return launch_k(gemm_ring_kernel<T, TN, TM, WGN, WGM, STAGES, EPI>, a, 2 * WGN * WGM * 64, (size_t)STAGES * (TN + TM) * 128, stream);
  } else {
    return -2;
  }
}
template <typename T, int TN, int TM, int WGN, int WGM, int STAGES>
int launch_ring(const GemmArgs& a, int epi, hipStream_t stream) {
  static_assert((size_t)STAGES * (TN + TM) * 128 <= 160 * 1024, "ring must fit the LDS");
  switch (epi) {
    case EPI_OP: return launch_ring_epi<T, TN, TM, WGN, WGM, STAGES, EPI_OP>(a, stream);
    case EPI_GELU_OP: return launch_ring_epi<T, TN, TM, WGN, WGM, STAGES, EPI_GELU_OP>(a, stream);
    case EPI_RES_F32: return launch_ring_epi<T, TN, TM, WGN, WGM, STAGES, EPI_RES_F32>(a, stream);
    case EPI_F32: return launch_ring_epi<T, TN, TM, WGN, WGM, STAGES, EPI_F32>(a, stream);
    case EPI_RES_LN:
      if constexpr (TN == 96) return launch_ring_epi<T, TN, TM, WGN, WGM, STAGES, EPI_RES_LN>(a, stream);   // the residual projections' tile only
      else return -2;
    default: return -2;
  }
}

#ifdef MRA_GEMM_EXPERIMENTS
#include "gemm_experiments.inc"
#endif

template <typename T>
int launch_t(const GemmArgs& a, int cfg, int epi, hipStream_t stream) {
  if (cfg == 8) return counted(GF_RING_144x128, epi, launch_ring<T, 144, 128, 1, 4, 4>(a, epi, stream));   // a wave: all 144 weight rows x 32 activation rows
  if (cfg == 9) return counted(GF_RING_192x128, epi, launch_ring<T, 192, 128, 2, 2, 4>(a, epi, stream));
  if (cfg == 10) return counted(GF_RING_96x64, epi, launch_ring<T, 96, 64, 2, 2, 7>(a, epi, stream));
  if (cfg == 3) return counted(GF_WS_128x384, epi, launch_ws_fold<T>(a, epi, stream));
  if (cfg == 4) return counted(GF_WS_176x384, epi, launch_ws_pv<T>(a, epi, stream));
  if (cfg == 5) return counted(GF_K128_64x128, epi, launch_k128<T, 64, 128, 2, 2>(a, epi, stream));   // 64 weight rows x 128 activation rows, 128-deep steps
  if (cfg == 6) return counted(GF_P8_TAIL, epi, launch_p8<T>(a, epi, stream, true));              // 128 weight rows x 512 activation rows, eight phases
  if (cfg == 7) {                                                        // N = 256 k + 128: full tiles + one 128 x 512 tail tile per pair of row tiles
    constexpr size_t ldst = 2 * (128 + 512) * 128;
    switch (epi) {
      case EPI_OP: return counted(GF_P8_MIXED, epi, launch_k(gemm_p8_mixed_kernel<T, EPI_OP>, a, 512, ldst, stream));   // the ViT's un-padded QKV (N = 4224)
      case EPI_RES_OP: return counted(GF_P8_MIXED, epi, launch_k(gemm_p8_mixed_kernel<T, EPI_RES_OP>, a, 512, ldst, stream));
      case EPI_RES_OP_STAT: return counted(GF_P8_MIXED, epi, launch_k(gemm_p8_mixed_kernel<T, EPI_RES_OP_STAT>, a, 512, ldst, stream));
      case EPI_RES_F32: return counted(GF_P8_MIXED, epi, launch_k(gemm_p8_mixed_kernel<T, EPI_RES_F32>, a, 512, ldst, stream));
      case EPI_RES_F32_STAT: return counted(GF_P8_MIXED, epi, launch_k(gemm_p8_mixed_kernel<T, EPI_RES_F32_STAT>, a, 512, ldst, stream));
      case EPI_F32: return counted(GF_P8_MIXED, epi, launch_k(gemm_p8_mixed_kernel<T, EPI_F32>, a, 512, ldst, stream));
      default: return -2;
    }
  }
#ifdef MRA_GEMM_EXPERIMENTS
  if (g_variant != 5 && g_variant != 1) {
    const int rc = launch_experiment<T>(a, cfg, epi, g_variant, stream);
    if (rc != -100) return rc;
  }
#endif
  if (g_variant != 1) {
    if (cfg == 2) {
      bool even = true;
      for (int g = 0; g < a.ngroups; ++g) even = even && (a.p[g].K / 64) % 2 == 0 && a.p[g].K >= 128;
      if (g_p8 && even) {
        const int rc = launch_p8<T>(a, epi, stream);
        if (rc != -100) return counted(GF_P8_256, epi, rc);
      }
      return counted(GF_WS_256, epi, launch_ws<T>(a, epi, stream));
    }
    if (cfg == 0) {
      // 128-deep steps pay on the 64x64 tile once the K loop is long (FFN down-projection, K = 3072:
      // 367 -> 460 TF/s); at K = 768 the launch is prologue/epilogue-bound and nothing changes
      bool k128 = true;
      for (int g = 0; g < a.ngroups; ++g) k128 = k128 && a.p[g].K % 128 == 0 && a.p[g].K >= 2048;
      if (k128) return counted(GF_K128_64x64, epi, launch_k128<T, 64, 64, 2, 2>(a, epi, stream));
    }
  }
  if (cfg == 2) return counted(GF_V1_256, epi, launch_v1<T, 256, 256, 2, 4>(a, epi, stream));
  if (cfg == 1) return counted(GF_V1_128, epi, launch_v1<T, 128, 128, 2, 2>(a, epi, stream));
  return counted(GF_V1_64, epi, launch_v1<T, 64, 64, 2, 2>(a, epi, stream));
}

constexpr int kTile[3] = {64, 128, 256};
constexpr int kTileN[11] = {64, 128, 256, 128, 176, 64, 128, 256, 144, 192, 96};   // weight rows per tile (config 7: 256, then 128 for the last column tile)
constexpr int kTileM[11] = {64, 128, 256, 384, 384, 128, 512, 256, 128, 128, 64};   // activation rows per tile (config 3: explicit only, GemmProb::tile_cfg = 4)

}  // namespace


long long gemm_launch_count(int family, int epi) {
  if (family < 0 || family >= GEMM_FAMILIES || epi < 0 || epi >= 16) return -1;
  return g_launches[family][epi].load(std::memory_order_relaxed);
}
#ifdef MRA_GEMM_EXPERIMENTS
void gemm_force_config(int cfg) { g_force_cfg = cfg; }
void gemm_force_variant(int v) { g_variant = v; }
void gemm_set_tile_order(int order) { g_order = order; }
void gemm_set_eight_phase(int on) { g_p8 = on; }
void gemm_set_debug_buffer(unsigned long long* p) { g_dbg = p; }
#endif

int gemm_pick_config(const GemmProb* probs, int ngroups) {
  if (g_force_cfg >= 0) return g_force_cfg;
  if (probs[0].tile_cfg > 0) return probs[0].tile_cfg - 1;
  // Largest tile that still gives every CU work: 256 CUs; aim for >= 2 waves of workgroups
  // with the 64/128 tiles and >= 1 full wave with the 256 tile.
  int best = 0;
  for (int c = 2; c >= 0; --c) {
    const int t = kTile[c];
    long long tiles = 0;
    bool ok = true;
    for (int g = 0; g < ngroups; ++g) {
      if (probs[g].N % t && !probs[g].n_ragged && !probs[g].n_mask) ok = false;
      tiles += (long long)((probs[g].M + t - 1) / t) * ((probs[g].N + t - 1) / t) * (probs[g].batch > 1 ? probs[g].batch : 1);
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
  if (ngroups < 1 || ngroups > GEMM_MAX_GROUPS) return -1;
  const int cfg = gemm_pick_config(probs, ngroups);
  if (cfg < 0 || cfg > 10) return -1;
  const int t = kTileN[cfg], tm = kTileM[cfg];
  GemmArgs a;
  a.ngroups = ngroups;
  int tiles = 0;
  for (int g = 0; g < ngroups; ++g) {
    a.p[g] = probs[g];
    GemmProb& p = a.p[g];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
    if (p.K % 64 || (p.N % t && !p.n_ragged && !p.n_mask && cfg != 7)) return -1;
    if (cfg == 5 && p.K % 128) return -1;
    if (cfg >= 8 && (p.n_ragged || p.n_mask || p.batch > 1 || p.w_ld || (epi != EPI_OP && epi != EPI_GELU_OP && epi != EPI_RES_F32 && epi != EPI_F32 && epi != EPI_RES_LN))) return -1;   // ring tiles: plain problems
    if (cfg >= 8 && (epi == EPI_RES_F32 || epi == EPI_F32 || epi == EPI_RES_LN) && ((p.c.ld & 3) || (p.c.item_stride & 3))) return -1;   // their outputs leave as 16-byte pieces
    if (epi == EPI_RES_LN && (cfg != 10 || !p.R || p.r.rpi <= 0 || !p.ln_gain || !p.ln_bias || !p.ln_counter || (!p.ln_y32 && !p.ln_y16) || p.N % 256 || p.N > 1024 ||
                              (p.ln_y32 && (p.ln_y32v.rpi <= 0 || (p.ln_y32v.ld & 3) || (p.ln_y32v.item_stride & 3))) ||
                              (p.ln_y16 && (p.ln_y16v.rpi <= 0 || (p.ln_y16v.ld & 3) || (p.ln_y16v.item_stride & 3))))) return -1;
    if (cfg == 6 && (p.K % 128 || p.N != 128 || p.n_mask || p.n_ragged || p.batch > 1)) return -1;   // the eight-phase tail tile: one column tile, even K steps
    if (cfg == 7 && (p.K % 128 || p.N % 256 != 128 || p.N < 384 || p.n_mask || p.n_ragged || p.batch > 1 || ngroups != 1)) return -1;
    if (p.n_mask && ((epi != EPI_RES_F32 && epi != EPI_F32 && epi != EPI_RES_OP && epi != EPI_OP && epi != EPI_GELU_OP) || (p.N & 3) || p.n_ragged)) return -1;
    if (p.n_mask && epi != EPI_RES_F32 && epi != EPI_F32 && (p.N & 63)) return -1;   // the 16-bit epilogues leave in 64-column blocks
    if (p.n_ragged && (p.bias || epi == EPI_KV || epi == EPI_RES_F32)) return -1;
    if (epi == EPI_SOFTPART && (cfg != 4 || !p.stat_m || !p.stat_l || p.bias || (p.c.ld & 3))) return -1;
    if (p.w_ld && (cfg != 4 || epi != EPI_OP || (p.w_ld & 7) || p.k_rows <= 0 || p.N % 176)) return -1;
    if (p.w_kwrap && (p.w_kwrap < 0 || p.w_ld || (cfg != 3 && cfg != 4) || 2 * p.w_kwrap * 64 != p.K)) return -1;   // 128 x 384 / 176 x 384 loader-wave tiles only: K = 2 passes over the weights
    if (p.pscale && (!p.w_ld || cfg != 4 || p.M > 384 || p.ps_ntiles <= 0 || p.K > p.ps_ntiles * 176 + 4 * 176)) return -1;   // M <= the 384-row tile (slices hold 512 rows)
    if (p.batch < 0) return -1;
    if (p.a.rpi <= 0 || (epi != EPI_KV && p.c.rpi <= 0)) return -1;
    if ((epi == EPI_RES_F32 || epi == EPI_RES_F32_STAT) && (!p.R || p.r.rpi <= 0)) return -1;
    if (epi == EPI_RES_F32_STAT || epi == EPI_RES_OP_STAT || epi == EPI_LNF_OP || epi == EPI_LNF_GELU_OP) {   // eight-phase tiles only, one plain problem
      bool even = (p.K / 64) % 2 == 0 && p.K >= 128;
      if (!(cfg == 2 && even && g_p8 && g_variant != 1) && !(cfg == 7 && (epi == EPI_RES_F32_STAT || epi == EPI_RES_OP_STAT))) return -1;
      if (ngroups != 1 || p.batch > 1 || p.n_mask || p.n_ragged || !p.ln_y32) return -1;
      if (epi == EPI_RES_OP_STAT && (p.N % 64 || !p.aux || (p.c.ld & 7) || (p.c.item_stride & 7))) return -1;
      if (epi == EPI_RES_F32_STAT && (p.N % 128 || !p.ln_y16 || p.ln_y16v.rpi <= 0 || (p.ln_y16v.ld & 7) || (p.ln_y16v.item_stride & 7) || (p.c.ld & 3) || (p.c.item_stride & 3))) return -1;
      if ((epi == EPI_LNF_OP || epi == EPI_LNF_GELU_OP) && (!p.ln_gain || (p.c.ld & 7) || (p.c.item_stride & 7))) return -1;
    }
    if (epi == EPI_RES_LN && cfg != 10) return -1;
    if (epi == EPI_KV && (p.kv_tokens <= 0 || p.kv_heads <= 0 || p.kv_items <= 0)) return -1;
    if ((epi == EPI_OP || epi == EPI_GELU_OP || epi == EPI_RES_OP) && ((p.c.ld & 7) || (p.c.item_stride & 7))) return -1;  // 16-byte stores
    if ((epi == EPI_RES_OP || epi == EPI_RES_OP_STAT) && (!p.aux || p.n_ragged)) return -1;
    if ((epi == EPI_GELU_BOTH || epi == EPI_GELU_BWD) && (!p.aux || (p.c.ld & 3) || (p.c.item_stride & 3) || p.n_ragged)) return -1;
    p.mtiles = (p.M + tm - 1) / tm;
    p.ntiles = (p.N + t - 1) / t;
    p.tile_begin = tiles;
    if (cfg == 7) tiles += (p.mtiles + 1) / 2 * (2 * (p.N >> 8) + 1);   // per pair of row tiles: 2 k full tiles + one tail tile
    else tiles += p.mtiles * p.ntiles * (p.batch > 1 ? p.batch : 1);
  }
  for (int g = ngroups; g < GEMM_MAX_GROUPS; ++g) a.p[g] = a.p[0];
  a.total_tiles = tiles;
  a.dbg = g_dbg;
  a.order = g_order ? g_order : probs[0].order;
  return op_dtype == OP_F16 ? launch_t<f16>(a, cfg, epi, stream) : launch_t<bf16>(a, cfg, epi, stream);
}

}  // namespace mra
