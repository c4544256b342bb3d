// EVA ViT-g/14 visual encoder (row A1 / N4): the callee of the reference's per-frame loop,
// `ln(encoder(frame))` at models/xinstructblip.py:262-266, built by create_eva_vit_g(224, 0, False, "fp16") (:658-666).
// LAVIS is absent offline; the arithmetic is the published EVA-CLIP-g geometry as LAVIS instantiates it (39 pre-LN blocks,
// width 1408, 16 heads x 88, MLP 6144, 14 x 14 patches of a 224 x 224 frame -> 257 tokens, q / v bias without k bias, exact
// GELU, LayerNorm eps 1e-6, no final norm) -- pinned to transformers' InstructBlipVisionModel (HF modeling_instructblip.py:
// 101-440) by mraudio_amd/models/eva_vit.py and tests/golden/vit_g.npz.
//
// One batched forward over ALL frames of a step (the reference calls the encoder T times at batch B):
//   im2col -> patch GEMM (+ bias + position embedding in the epilogue) -> [CLS] rows
//   39 x { LN -> QKV GEMM -> attention core -> projection GEMM + residual -> LN -> fc1 GEMM + GELU -> fc2 GEMM + residual }
// 97 % of the flops are the four GEMMs per block at M = frames x 257 rows: the eight-phase 256 x 256 kernel of gemm.hip (QKV, fc1;
// the N = 1408 ones as full tiles plus a 128 x 512 tail tile per pair of row tiles, GemmProb::tile_cfg 8) with bias / GELU /
// residual fused: the residual GEMMs START their accumulators at bias + residual, so their epilogue only stores.  The fp32
// residual stream is updated in place and IS the output.  LayerNorms: folded into the GEMMs on either side (default with the fp32 stream,
// see vit_fold_weight_kernel below) or separate launches that write the f16 operand of the next GEMM (f16 stream; ln_fold 0).
// Head dimension 88 is not a multiple of the MFMA K step: the QKV weight is regrouped [q|k|v][head][96] with eight zero rows
// per head, so Q, K, V come out of the GEMM padded to 96 and the attention core (vit_attn_kernel below) runs on 3 x 32-deep
// MFMA steps; the zero columns add nothing to any dot product.  (Round 3 measured the alternative -- Q | K | V un-padded out of the
// GEMM, N = 4224 on the mixed eight-phase kernel, heads zero-extended on the core's LDS fill: the GEMM gains 72 us per 256 frames,
// the core loses 54 on its 176-byte head rows, and over 1024 frames the encoder ran 575.5-576.7 against 571.7 ms in the same
// session: not kept.  A run-time head stride in the core alone cost 30 % of it.)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mra_common.h"
#include "mra_handle.h"

using namespace mra;
using namespace mra_host;

namespace {

constexpr int HD_PAD = 96;      // padded head dimension
constexpr int KS_PAD = 288;     // keys padded to 9 MFMA K steps of 32 (257 valid)
constexpr int KV_PITCH = 208;   // LDS row pitch of K / V tiles: 13 sixteen-byte slots -> conflict-free ds_read_b128 over 16 rows
constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------------------
// attention core: one workgroup of nine waves per (frame, head); K and V of the head are staged once in LDS and each wave
// takes 16-query blocks (17 blocks: at most two per wave).  S^T = K Q^T puts the keys on the MFMA row index, so a lane
// holds, for its query, keys 16 i + 4 c + 0..3 of every 16-key fragment i (c = lane >> 4): the softmax is a reduction over
// the lane's own registers plus two lane swaps, and -- since the order of the contraction index inside an MFMA is free as
// long as both operands agree -- fragments 2 ks and 2 ks + 1 together ARE the B operand of K step ks of O^T = V^T P^T
// (keys 32 ks + 4 c + 0..3 and 32 ks + 16 + 4 c + 0..3): P never leaves the registers.  V^T takes the same key order from
// two transposed LDS reads (ds_read_b64_tr_b16) of V [key][d].
// ---------------------------------------------------------------------------------------------------------
constexpr int ATT_WAVES = 9;
// The softmax is VALU-bound next to 108 MFMAs per 16-query block (72 scores per lane), so it is kept lean: the scale rides the
// exponent's fma; with the sequence length known at compile time (SC = 257 for ViT-g/224) only key fragment 16 is masked and
// fragment 17 (all padding) is neither multiplied nor exponentiated; and the row sum comes out of the P V MFMAs themselves:
// V's first padding column (d = hd) is set to one when the head is staged, so O^T[hd][query] = sum of the ROUNDED P row.
template <typename T, int SC>
__global__ void __launch_bounds__(ATT_WAVES * 64) vit_attn_kernel(const T* qkv, T* ctx, int S_rt, int heads, int hd, float sl2) {
  const int S = SC ? SC : S_rt;
  const bool ones = hd < HD_PAD;    // a padding column exists: the row sum rides the MFMAs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + KS_PAD * KV_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frame = blockIdx.x / heads, head = blockIdx.x - frame * heads;
  const int ld = 3 * heads * HD_PAD;
  const T* base = qkv + (long long)frame * S * ld + head * HD_PAD;
  using V8 = typename Vec8<T>::type;
  // ---- K, V rows of this head -> LDS (rows past S zeroed).  Every load is issued before the first LDS write and none sits
  // under a lane-dependent branch (a clamped row is loaded and zeroed instead): one round trip, not one per chunk.
  {
    constexpr int NCH = (KS_PAD * 12 + ATT_WAVES * 64 - 1) / (ATT_WAVES * 64);
    V8 kr[NCH], vr[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int c = min(tid + u * ATT_WAVES * 64, KS_PAD * 12 - 1);
      const int row = c / 12, ch = c - row * 12;
      const T* p = base + (long long)min(row, S - 1) * ld + ch * 8;
      kr[u] = *reinterpret_cast<const V8*>(p + heads * HD_PAD);
      vr[u] = *reinterpret_cast<const V8*>(p + 2 * heads * HD_PAD);
    }
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int c = tid + u * ATT_WAVES * 64;
      const int row = c / 12, ch = c - row * 12;
      if (row >= S) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { kr[u][e] = from_f32<T>(0.f); vr[u][e] = kr[u][e]; }
      }
      if (ones && ch == (hd >> 3)) {   // the ones column (masked keys carry P = 0)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e == (hd & 7)) vr[u][e] = from_f32<T>(1.f);
      }
      if (c < KS_PAD * 12) {
        *reinterpret_cast<V8*>(Ks + row * KV_PITCH + ch * 16) = kr[u];
        *reinterpret_cast<V8*>(Vs + row * KV_PITCH + ch * 16) = vr[u];
      }
    }
  }
  __syncthreads();
  const int lm = lane & 15, lc = lane >> 4;
  const int nblocks = (S + 15) >> 4;
  for (int qb = wave; qb < nblocks; qb += ATT_WAVES) {
    const int q0 = qb * 16;
    // Q fragments (B operand): lane (query lm, chunk lc) holds Q[q0 + lm][32 ks + 8 lc .. + 7]
    V8 qf[3];
    {
      const T* qp = base + (long long)min(q0 + lm, S - 1) * ld + 8 * lc;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 32 * ks);
    }
    constexpr int NFR = SC ? (SC + 15) / 16 : KS_PAD / 16;   // key fragments with at least one valid key (compile-time S)
    f32x4 sc[KS_PAD / 16];
    float mx = -3.0e38f;     // of the raw scores: sl2 > 0
#pragma unroll
    for (int i = 0; i < KS_PAD / 16; ++i) {
      if (i >= NFR) { sc[i] = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}; continue; }
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const V8 kf = *reinterpret_cast<const V8*>(Ks + (16 * i + lm) * KV_PITCH + (4 * ks + lc) * 16);
        a = mfma16<T>(kf, qf[ks], a);
      }
      if (SC && 16 * i + 16 <= SC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, a[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (16 * i + 4 * lc + e >= S) a[e] = -3.0e38f;
          mx = fmaxf(mx, a[e]);
        }
      }
      sc[i] = a;
    }
    // Scheduling of the K-fragment reads against the MFMAs: six reads run ahead, then one read per MFMA.  Left alone
    // the scheduler hoists every read above the first MFMA and the allocator spills (644 bytes of scratch per lane, 0.96 ms per
    // layer); strictly one read per MFMA exposes the LDS latency on every MFMA (0.37 ms).
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 3 * NFR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mxs = mx * sl2;
    float l = 0.f;
    // O^T[d][query] = sum_keys V^T[d][key] P^T[key][query]
    f32x4 ot[HD_PAD / 16];
#pragma unroll
    for (int df = 0; df < HD_PAD / 16; ++df) ot[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS_PAD / 32; ++ks) {
      V8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * ks][e], sl2, -mxs)));          // masked keys: exp2(-huge) = 0
        pf[4 + e] = 2 * ks + 1 < NFR ? from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * ks + 1][e], sl2, -mxs))) : from_f32<T>(0.f);
        if (!ones) l += (float)pf[e] + (float)pf[4 + e];
      }
      // V^T fragment of d block df in the same key order: transposed 4 x 16 blocks at keys 32 ks + 4 lc and 32 ks + 16 + 4 lc
      const char* vb = Vs + (32 * ks + 4 * lc + (lm >> 2)) * KV_PITCH + (lane & 3) * 8;
#pragma unroll
      for (int df = 0; df < HD_PAD / 16; ++df) {
        const i16x4 c0 = lds_read_tr4(vb + df * 32), c1 = lds_read_tr4(vb + df * 32 + 16 * KV_PITCH);
        i16x8 v;
        v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3]; v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
        ot[df] = mfma16<T>(__builtin_bit_cast(V8, v), pf, ot[df]);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // same for the 108 transposed V reads: four fragments ahead
#pragma unroll
    for (int i = 0; i < (HD_PAD / 16) * (KS_PAD / 32); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    if (ones) {
      // O^T[hd][query]: fragment hd / 16, row hd % 16 = 4 c + e
      float lo = 0.f;
#pragma unroll
      for (int df = 0; df < HD_PAD / 16; ++df)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (df == (hd >> 4) && e == (hd & 3)) lo = ot[df][e];
      l = __shfl(lo, ((hd & 15) >> 2) * 16 + lm);
    } else {
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
    }
    const float inv = 1.0f / l;
    if (q0 + lm < S) {
      T* crow = ctx + ((long long)frame * S + q0 + lm) * (heads * hd) + head * hd;
#pragma unroll
      for (int df = 0; df < HD_PAD / 16; ++df) {
        const int d = 16 * df + 4 * lc;
        if (d + 3 < hd) {
          typename Vec4<T>::type o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(ot[df][e] * inv);
          *reinterpret_cast<typename Vec4<T>::type*>(crow + d) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same core as a PERSISTENT kernel (S = 257): one workgroup of eight waves per CU walks over (frame, head) units; while the waves work
// on unit u out of LDS, the K / V rows of unit u + gridDim.x are already on their way into registers (56 per lane) and go to LDS behind a
// barrier when u is done.  With one workgroup per CU (the head's K and V fill 120 KB of LDS) the stand-alone kernel above cannot hide that fetch:
// 99 KB per unit at the ~21 GB/s one CU gets from HBM is several microseconds of a ~17 us unit.  The Q fragments of a wave's (up to three)
// query blocks are requested BEFORE the prefetch: vmcnt retires in order, and a Q load issued behind it would wait for all of it.
// Bit-identical to the kernel above -- and measured slower (303 vs 277 us per launch): 249 registers per lane allow eight waves, and 17 query
// blocks over eight waves leave one wave with three blocks in series where nine waves have at most two.  Opt-in (mra_vit_set_option "attn_persist").
// ---------------------------------------------------------------------------------------------------------
constexpr int ATTP_WAVES = 8;
template <typename T, int SC>
__global__ void __launch_bounds__(ATTP_WAVES * 64) vit_attn_persist_kernel(const T* qkv, T* ctx, int units, int heads, int hd, float sl2) {
  static_assert(SC > 0, "compile-time sequence length");
  constexpr int S = SC;
  const bool ones = hd < HD_PAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + KS_PAD * KV_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * heads * HD_PAD;
  using V8 = typename Vec8<T>::type;
  constexpr int NT = ATTP_WAVES * 64;
  constexpr int NCH = (KS_PAD * 12 + NT - 1) / NT;
  constexpr int NBLK = (SC + 15) / 16, BPW = (NBLK + ATTP_WAVES - 1) / ATTP_WAVES;   // query blocks; per wave at most
  auto unit_base = [&](int u) {
    const int frame = u / heads, head = u - frame * heads;
    return qkv + (long long)frame * S * ld + head * HD_PAD;
  };
  V8 kr[NCH], vr[NCH];
  auto fetch = [&](const T* base) {     // every load issued back to back, none under a lane-dependent branch (a clamped row is loaded and zeroed later)
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int c = min(tid + u * NT, KS_PAD * 12 - 1);
      const int row = c / 12, ch = c - row * 12;
      const T* p = base + (long long)min(row, S - 1) * ld + ch * 8;
      kr[u] = *reinterpret_cast<const V8*>(p + heads * HD_PAD);
      vr[u] = *reinterpret_cast<const V8*>(p + 2 * heads * HD_PAD);
    }
  };
  auto to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int c = tid + u * NT;
      const int row = c / 12, ch = c - row * 12;
      if (row >= S) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { kr[u][e] = from_f32<T>(0.f); vr[u][e] = kr[u][e]; }
      }
      if (ones && ch == (hd >> 3)) {   // the ones column (masked keys carry P = 0)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e == (hd & 7)) vr[u][e] = from_f32<T>(1.f);
      }
      if (c < KS_PAD * 12) {
        *reinterpret_cast<V8*>(Ks + row * KV_PITCH + ch * 16) = kr[u];
        *reinterpret_cast<V8*>(Vs + row * KV_PITCH + ch * 16) = vr[u];
      }
    }
  };
  const int lm = lane & 15, lc = lane >> 4;
  int u = blockIdx.x;
  if (u >= units) return;
  fetch(unit_base(u));
  to_lds();
  __syncthreads();
  for (; u < units; u += gridDim.x) {
    const T* base = unit_base(u);
    const int frame = u / heads, head = u - frame * heads;
    const int un = u + gridDim.x;
    // Q fragments (B operand) of this wave's blocks: lane (query lm, chunk lc) holds Q[q0 + lm][32 ks + 8 lc .. + 7]
    // (assembly loads: the compiler would sink ordinary loads to their first use -- behind the prefetch -- and close them with vmcnt(0))
    V8 qfs[BPW][3];
    static_assert(BPW == 3, "the counted wait below names nine registers");
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      const int q0 = min(wave + b * ATTP_WAVES, NBLK - 1) * 16;
      const T* qp = base + (long long)min(q0 + lm, S - 1) * ld + 8 * lc;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qfs[b][ks]) : "v"(qp + 32 * ks) : "memory");
    }
    // Unconditional (the last unit of a workgroup fetches itself again and drops it): under a branch the compiler must assume the loads may not
    // have been issued and counts the Q fragments' vmcnt without them -- i.e. waits for most of the prefetch before the first MFMA.
    fetch(unit_base(min(un, units - 1)));
    // the Q fragments have landed once at most the 2 NCH prefetch loads behind them are outstanding
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(qfs[0][0]), "+v"(qfs[0][1]), "+v"(qfs[0][2]), "+v"(qfs[1][0]), "+v"(qfs[1][1]), "+v"(qfs[1][2]), "+v"(qfs[2][0]), "+v"(qfs[2][1]), "+v"(qfs[2][2])
                 : "n"(2 * NCH)
                 : "memory");
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      const int qb = wave + b * ATTP_WAVES;
      if (qb >= NBLK) break;                        // wave-uniform
      const int q0 = qb * 16;
      V8 qf[3] = {qfs[b][0], qfs[b][1], qfs[b][2]};
    constexpr int NFR = SC ? (SC + 15) / 16 : KS_PAD / 16;   // key fragments with at least one valid key (compile-time S)
    f32x4 sc[KS_PAD / 16];
    float mx = -3.0e38f;     // of the raw scores: sl2 > 0
#pragma unroll
    for (int i = 0; i < KS_PAD / 16; ++i) {
      if (i >= NFR) { sc[i] = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}; continue; }
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const V8 kf = *reinterpret_cast<const V8*>(Ks + (16 * i + lm) * KV_PITCH + (4 * ks + lc) * 16);
        a = mfma16<T>(kf, qf[ks], a);
      }
      if (SC && 16 * i + 16 <= SC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, a[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (16 * i + 4 * lc + e >= S) a[e] = -3.0e38f;
          mx = fmaxf(mx, a[e]);
        }
      }
      sc[i] = a;
    }
    // Scheduling of the K-fragment reads against the MFMAs: six reads run ahead, then one read per MFMA.  Left alone
    // the scheduler hoists every read above the first MFMA and the allocator spills (644 bytes of scratch per lane, 0.96 ms per
    // layer); strictly one read per MFMA exposes the LDS latency on every MFMA (0.37 ms).
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 3 * NFR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mxs = mx * sl2;
    float l = 0.f;
    // O^T[d][query] = sum_keys V^T[d][key] P^T[key][query]
    f32x4 ot[HD_PAD / 16];
#pragma unroll
    for (int df = 0; df < HD_PAD / 16; ++df) ot[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS_PAD / 32; ++ks) {
      V8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * ks][e], sl2, -mxs)));          // masked keys: exp2(-huge) = 0
        pf[4 + e] = 2 * ks + 1 < NFR ? from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * ks + 1][e], sl2, -mxs))) : from_f32<T>(0.f);
        if (!ones) l += (float)pf[e] + (float)pf[4 + e];
      }
      // V^T fragment of d block df in the same key order: transposed 4 x 16 blocks at keys 32 ks + 4 lc and 32 ks + 16 + 4 lc
      const char* vb = Vs + (32 * ks + 4 * lc + (lm >> 2)) * KV_PITCH + (lane & 3) * 8;
#pragma unroll
      for (int df = 0; df < HD_PAD / 16; ++df) {
        const i16x4 c0 = lds_read_tr4(vb + df * 32), c1 = lds_read_tr4(vb + df * 32 + 16 * KV_PITCH);
        i16x8 v;
        v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3]; v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
        ot[df] = mfma16<T>(__builtin_bit_cast(V8, v), pf, ot[df]);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // same for the 108 transposed V reads: four fragments ahead
#pragma unroll
    for (int i = 0; i < (HD_PAD / 16) * (KS_PAD / 32); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    if (ones) {
      // O^T[hd][query]: fragment hd / 16, row hd % 16 = 4 c + e
      float lo = 0.f;
#pragma unroll
      for (int df = 0; df < HD_PAD / 16; ++df)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (df == (hd >> 4) && e == (hd & 3)) lo = ot[df][e];
      l = __shfl(lo, ((hd & 15) >> 2) * 16 + lm);
    } else {
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
    }
    const float inv = 1.0f / l;
    if (q0 + lm < S) {
      T* crow = ctx + ((long long)frame * S + q0 + lm) * (heads * hd) + head * hd;
#pragma unroll
      for (int df = 0; df < HD_PAD / 16; ++df) {
        const int d = 16 * df + 4 * lc;
        if (d + 3 < hd) {
          typename Vec4<T>::type o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(ot[df][e] * inv);
          *reinterpret_cast<typename Vec4<T>::type*>(crow + d) = o;
        }
      }
    }
    }
    __syncthreads();                                // every wave is done with the K / V of unit u
    if (un < units) to_lds();
    __syncthreads();
  }
}

// frames [n][3][img][img] (f32 or f16) -> patches [n * np * np][kpad] in the operand dtype, k = (c * ps + i) * ps + j, zeros past 3 ps^2
template <typename TI, typename T>
__global__ void __launch_bounds__(256) vit_im2col_kernel(const TI* x, T* out, long long total, int img, int ps, int np, int kpad) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % kpad);
  const long long patch = idx / kpad;
  float v = 0.f;
  if (k < 3 * ps * ps) {
    const int c = k / (ps * ps), r = k - c * ps * ps, i = r / ps, j = r - i * ps;
    const int px = (int)(patch % np), py = (int)((patch / np) % np);
    const long long f = patch / (np * np);
    v = (float)x[((f * 3 + c) * img + py * ps + i) * img + px * ps + j];
  }
  out[idx] = from_f32<T>(v);
}

// x[frame][0][:] = cls + pos[0]
__global__ void __launch_bounds__(256) vit_cls_kernel(float* x, const float* cls, const float* pos, int frames, int S, int D) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)frames * D) return;
  const int c = (int)(idx % D);
  x[(idx / D) * S * D + c] = cls[c] + pos[c];
}

// dst (operand dtype) [rows_out][cols_out] <- src [*][cols_in]: row r takes source row map(r) (or zeros), columns past cols_in zero.
// mode 0: identity rows (column padding only); mode 1: QKV regroup, r = (s * heads + h) * 96 + d <- s * heads * hd + h * hd + d for d < hd
template <typename TI, typename T>
__global__ void __launch_bounds__(256) vit_pack_kernel(const TI* src, T* dst, long long total, int cols_in, int cols_out, int mode, int heads, int hd) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cols_out);
  const long long r = idx / cols_out;
  long long sr = r;
  if (mode == 1) {
    const int d = (int)(r % HD_PAD);
    const long long sh = r / HD_PAD;
    sr = d < hd ? sh * hd + d : -1;
  }
  dst[idx] = from_f32<T>(sr >= 0 && c < cols_in ? (float)src[sr * cols_in + c] : 0.f);
}

template <typename TI>
int pack_to(const void* src, void* dst, int op, long long rows_out, int cols_in, int cols_out, int mode, int heads, int hd, hipStream_t st) {
  const long long total = rows_out * cols_out;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (op == OP_F16) hipLaunchKernelGGL((vit_pack_kernel<TI, f16>), grid, block, 0, st, (const TI*)src, (f16*)dst, total, cols_in, cols_out, mode, heads, hd);
  else hipLaunchKernelGGL((vit_pack_kernel<TI, bf16>), grid, block, 0, st, (const TI*)src, (bf16*)dst, total, cols_in, cols_out, mode, heads, hd);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs around it (fp32 residual stream; kernels.h, EPI_RES_F32_STAT / EPI_LNF_*).  LN(x) W^T + b =
// rstd (x W'^T - mu colsum(W')) + (b + W beta) with W' = W diag(gain): the QKV and fc1 GEMMs run on the op-dtype copy of the RAW rows that
// the residual GEMM before them wrote next to its fp32 store, and finish the LayerNorm in their epilogue from (mu, rstd) per row.  What is
// gone per block: two reads of the fp32 rows and two launches (94 us each at 256 frames: 5 % of the encoder).
// ---------------------------------------------------------------------------------------------------------
// W' = W diag(gain) in the operand dtype, cs[n] = sum_k W'[n][k] (of the ROUNDED W': what the MFMAs multiply), bf[n] = bias[n] + sum_k beta[k] W[n][k]
template <typename T>
__global__ void __launch_bounds__(256) vit_fold_weight_kernel(const T* W, const float* gain, const float* beta, const float* bias, T* Wf, float* cs,
                                                              float* bf, int N, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  const T* w = W + (long long)n * K;
  T* wf = Wf + (long long)n * K;
  float c = 0.f, b = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float wv = (float)w[k];
    const T f = from_f32<T>(wv * gain[k]);
    wf[k] = f;
    c += (float)f;
    b = __builtin_fmaf(beta[k], wv, b);
  }
  c = wave_sum(c);
  b = wave_sum(b);
  if (lane == 0) { cs[n] = c; bf[n] = b + (bias ? bias[n] : 0.f); }
}

// groups[m][g] = (mean, sum of squared deviations) of `gsz` columns -> stats[m] = (mean, rstd) of the row (Chan et al.: equal counts)
__global__ void __launch_bounds__(256) vit_group_stats_kernel(const float2* groups, long long M, int G, float gsz, float eps, float2* stats) {
  const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float2* g = groups + m * G;
  float mean = 0.f;
  for (int i = 0; i < G; ++i) mean += g[i].x;
  mean /= (float)G;
  float m2 = 0.f;
  for (int i = 0; i < G; ++i) { const float d = g[i].x - mean; m2 += g[i].y + gsz * d * d; }
  stats[m] = float2{mean, 1.0f / sqrtf(m2 / (gsz * (float)G) + eps)};
}

// the same for a residual stream kept in the operand dtype (the stream is its own copy): statistics of the rounded rows
template <typename T>
__global__ void __launch_bounds__(256) vit_row_stats16_kernel(const T* x, long long M, int D, float eps, float2* stats) {
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const T* row = x + m * D;
  const int it = D >> 7;
  float2 v[16];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    v[t] = t < it ? float2{(float)row[(t * 64 + lane) * 2], (float)row[(t * 64 + lane) * 2 + 1]} : float2{0.f, 0.f};
    s += v[t].x + v[t].y;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < it) { const float a = v[t].x - mean, b = v[t].y - mean; q += a * a + b * b; }
  q = wave_sum(q);
  if (lane == 0) stats[m] = float2{mean, 1.0f / sqrtf(q / (float)D + eps)};
}

// the rows the first block reads (patch embedding + positions): (mean, rstd) and the op-dtype copy; one wave per row, D = 128 k <= 2048
template <typename T>
__global__ void __launch_bounds__(256) vit_row_stats_kernel(const float* x, long long M, int D, float eps, float2* stats, T* x16) {
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float2* row = reinterpret_cast<const float2*>(x + m * D);
  const int it = D >> 7;
  float2 v[16];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    v[t] = t < it ? row[t * 64 + lane] : float2{0.f, 0.f};
    s += v[t].x + v[t].y;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < it) { const float a = v[t].x - mean, b = v[t].y - mean; q += a * a + b * b; }
  q = wave_sum(q);
  if (lane == 0) stats[m] = float2{mean, 1.0f / sqrtf(q / (float)D + eps)};
  T* o = x16 + m * D;
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < it) { o[(t * 64 + lane) * 2] = from_f32<T>(v[t].x); o[(t * 64 + lane) * 2 + 1] = from_f32<T>(v[t].y); }
}

struct VitLayer {
  float *n1g, *n1b, *n2g, *n2b, *bqkv, *bproj, *bfc1, *bfc2;
  void *wqkv, *wproj, *wfc1, *wfc2;
  // LayerNorm folded in (prepared by fold_weights): W diag(gain), its row sums, bias + W beta
  void *wqkv_f, *wfc1_f;
  float *cs_qkv, *bf_qkv, *cs_fc1, *bf_fc1;
};

}  // namespace

struct mra_vit {
  mra_vit_cfg cfg;
  int device = 0;
  int S = 0, np = 0, kpad = 0, nqkv = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::map<std::string, int> loaded;   // name -> 1 once loaded
  float *cls = nullptr, *pos = nullptr, *bpatch = nullptr;
  void* wpatch = nullptr;
  std::vector<VitLayer> layers;
  bool tail_tile = true;   // N = dim GEMMs: full 256-wide tiles + a 128 x 512 tail tile per pair of row tiles (false: a masked sixth 256-wide column tile)
  int proj_tile = 3;   // GemmProb::tile_cfg of the N = dim GEMMs: 256 x 256 with a masked last column tile (1408 = 5.5 tiles); the exact-fit
                       // 176 x 384 tile (tile_cfg 5) measured 1 % slower (615 vs 609 ms per 1024 frames): both sit on the fp32 epilogue
  // LayerNorms folded into the QKV / fc1 GEMMs (dim = 256 k + 128): mra_vit_set_option("ln_fold", 0 / 1)
  int ln_fold = 1;
  int gemm_persist = 1;   // QKV / fc1 on the eight-phase kernel as one persistent workgroup per CU (GemmProb::persist): 0 never, 1 always, 2 up to 64 rounds of
                          // the chip.  Measured in the encoder (r03final, alternating in one process, bit-identical outputs): 20.1 -> 19.9 ms at 32 frames,
                          // 139.8 -> 138.4 at 256, 547.4 -> 538.7 at 1024.  (The projection / fc2 GEMMs on the mixed-tile kernel lose with the same treatment:
                          // 549.4 against 538.9 ms -- their two tile kinds and the residual read at the head of a tile want the dispatcher's dynamic order; not built in.)
  int attn_persist = 0;   // S = 257: the attention core as one persistent workgroup per CU that prefetches the next (frame, head) unit (vit_attn_persist_kernel).
                          // Measured SLOWER (r03z: 303 vs 277 us per launch at 256 frames, 576-579 vs 573-576 ms per 1024 frames): its register budget allows
                          // eight waves, and 17 query blocks over eight waves put three blocks in series on one wave; opt-in
  int cus = 0;
  bool fold_ready = false;   // W diag(gain) etc. are up to date with the loaded parameters
  int op() const { return cfg.op_dtype == MRA_BF16 ? OP_BF16 : OP_F16; }
  bool can_fold() const {   // either residual dtype
    return proj_tile == 3 && tail_tile && cfg.dim % 256 == 128 && cfg.dim > 128 && cfg.dim <= 2048 && cfg.mlp % 128 == 0;
  }
};

namespace {

size_t vit_layout(mra_vit* h, char* base) {
  const mra_vit_cfg& c = h->cfg;
  const size_t D = c.dim, I = c.mlp;
  Carver cv(base);
  h->cls = cv.take<float>(D);
  h->pos = cv.take<float>((size_t)h->S * D);
  h->wpatch = cv.take<char>(D * h->kpad, 2);
  h->bpatch = cv.take<float>(D);
  h->layers.assign(c.depth, VitLayer{});
  for (auto& L : h->layers) {
    L.n1g = cv.take<float>(D); L.n1b = cv.take<float>(D); L.n2g = cv.take<float>(D); L.n2b = cv.take<float>(D);
    L.wqkv = cv.take<char>((size_t)h->nqkv * D, 2); L.bqkv = cv.take<float>(h->nqkv);
    L.wproj = cv.take<char>(D * D, 2); L.bproj = cv.take<float>(D);
    L.wfc1 = cv.take<char>(I * D, 2); L.bfc1 = cv.take<float>(I);
    L.wfc2 = cv.take<char>(D * I, 2); L.bfc2 = cv.take<float>(D);
    L.wqkv_f = cv.take<char>((size_t)h->nqkv * D, 2); L.cs_qkv = cv.take<float>(h->nqkv); L.bf_qkv = cv.take<float>(h->nqkv);
    L.wfc1_f = cv.take<char>(I * D, 2); L.cs_fc1 = cv.take<float>(I); L.bf_fc1 = cv.take<float>(I);
  }
  return cv.off;
}

int fold_weights(mra_vit* h, hipStream_t st) {
  const mra_vit_cfg& c = h->cfg;
  const int D = c.dim, I = c.mlp;
  for (auto& L : h->layers) {
    if (h->op() == OP_F16) {
      hipLaunchKernelGGL(vit_fold_weight_kernel<f16>, dim3((h->nqkv + 3) / 4), dim3(256), 0, st, (const f16*)L.wqkv, L.n1g, L.n1b, L.bqkv, (f16*)L.wqkv_f, L.cs_qkv, L.bf_qkv, h->nqkv, D);
      hipLaunchKernelGGL(vit_fold_weight_kernel<f16>, dim3((I + 3) / 4), dim3(256), 0, st, (const f16*)L.wfc1, L.n2g, L.n2b, L.bfc1, (f16*)L.wfc1_f, L.cs_fc1, L.bf_fc1, I, D);
    } else {
      hipLaunchKernelGGL(vit_fold_weight_kernel<bf16>, dim3((h->nqkv + 3) / 4), dim3(256), 0, st, (const bf16*)L.wqkv, L.n1g, L.n1b, L.bqkv, (bf16*)L.wqkv_f, L.cs_qkv, L.bf_qkv, h->nqkv, D);
      hipLaunchKernelGGL(vit_fold_weight_kernel<bf16>, dim3((I + 3) / 4), dim3(256), 0, st, (const bf16*)L.wfc1, L.n2g, L.n2b, L.bfc1, (bf16*)L.wfc1_f, L.cs_fc1, L.bf_fc1, I, D);
    }
  }
  if (hipGetLastError() != hipSuccess) return -4;
  h->fold_ready = true;
  return 0;
}

int copy_f32(const void* src, int dtype, float* dst, long long n, hipStream_t st) {
  return launch_convert(src, dtype, dst, MRA_F32, n, st);
}

}  // namespace

extern "C" {

void mra_vit_cfg_default(mra_vit_cfg* c) {
  c->dim = 1408; c->heads = 16; c->mlp = 6144; c->depth = 39; c->patch = 14; c->img = 224; c->ln_eps = 1e-6f; c->op_dtype = MRA_F16; c->residual_dtype = MRA_F32;
}

int mra_vit_create(const mra_vit_cfg* cfg, mra_vit** out) {
  if (!cfg || !out) return fail(MRA_EINVAL, "null argument");
  const mra_vit_cfg& c = *cfg;
  if (c.dim <= 0 || c.dim % 176 || c.dim % 64) return fail(MRA_EINVAL, "dim must be a multiple of 176 and of 64 (1408)");
  if (c.heads <= 0 || c.dim % c.heads || c.dim / c.heads > HD_PAD || (c.dim / c.heads) % 8) return fail(MRA_EINVAL, "head dimension must be a multiple of 8, <= 96");
  if ((3 * c.heads * HD_PAD) % 256 || c.mlp <= 0 || c.mlp % 256) return fail(MRA_EINVAL, "3 * heads * 96 and mlp must be multiples of 256");
  if (c.patch <= 0 || c.img <= 0 || c.img % c.patch || c.depth <= 0) return fail(MRA_EINVAL, "bad patch / image size / depth");
  if (c.op_dtype != MRA_F16 && c.op_dtype != MRA_BF16) return fail(MRA_EINVAL, "op_dtype must be MRA_F16 or MRA_BF16");
  if (c.residual_dtype != MRA_F32 && c.residual_dtype != c.op_dtype) return fail(MRA_EINVAL, "residual_dtype must be MRA_F32 or the operand dtype");
  mra_vit* h = new mra_vit();
  h->cfg = c;
  h->np = c.img / c.patch;
  h->S = h->np * h->np + 1;
  if (h->S > KS_PAD) { delete h; return fail(MRA_EINVAL, "more than 288 tokens per frame are not supported"); }
  h->kpad = (3 * c.patch * c.patch + 63) / 64 * 64;
  h->nqkv = 3 * c.heads * HD_PAD;
  HIP_TRY(hipGetDevice(&h->device));
  HIP_TRY(hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device));
  h->arena_bytes = vit_layout(h, nullptr);
  const hipError_t e = hipMalloc((void**)&h->arena, h->arena_bytes);
  if (e != hipSuccess) { delete h; return fail(MRA_ENOMEM, std::string("hipMalloc of the ViT parameter arena: ") + hipGetErrorString(e)); }
  vit_layout(h, h->arena);
  HIP_TRY(hipMemsetAsync(h->arena, 0, h->arena_bytes, 0));   // the key-bias third of bqkv and all padding stay zero
  *out = h;
  return MRA_OK;
}

void mra_vit_destroy(mra_vit* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  delete h;
}

int mra_vit_load(mra_vit* h, const char* name, const void* src, int32_t dtype, const int64_t* shape, int32_t ndim, void* stream_) {
  if (!h || !name || !src || (ndim > 0 && !shape)) return fail(MRA_EINVAL, "null argument");
  if (dtype < MRA_F32 || dtype > MRA_BF16) return fail(MRA_EINVAL, "bad dtype");
  const mra_vit_cfg& c = h->cfg;
  const long long D = c.dim, I = c.mlp, hd = c.dim / c.heads;
  long long numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= shape[i];
  hipStream_t st = as_stream(stream_);
  const std::string key(name);
  auto expect = [&](long long n) { return numel == n ? 0 : fail(MRA_EINVAL, "parameter " + key + ": expected " + std::to_string(n) + " elements, got " + std::to_string(numel)); };
  auto pack = [&](void* dst, long long rows_out, int cols_in, int cols_out, int mode) {
    switch (dtype) {
      case MRA_F32: return pack_to<float>(src, dst, h->op(), rows_out, cols_in, cols_out, mode, c.heads, (int)hd, st);
      case MRA_F16: return pack_to<f16>(src, dst, h->op(), rows_out, cols_in, cols_out, mode, c.heads, (int)hd, st);
      default: return pack_to<bf16>(src, dst, h->op(), rows_out, cols_in, cols_out, mode, c.heads, (int)hd, st);
    }
  };
  int rc = 0;
  if (key == "cls_token") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, h->cls, D, st); }
  else if (key == "pos_embed") { if ((rc = expect((long long)h->S * D))) return rc; rc = copy_f32(src, dtype, h->pos, (long long)h->S * D, st); }
  else if (key == "patch_embed.weight") { if ((rc = expect(D * 3 * c.patch * c.patch))) return rc; rc = pack(h->wpatch, D, 3 * c.patch * c.patch, h->kpad, 0); }
  else if (key == "patch_embed.bias") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, h->bpatch, D, st); }
  else if (key.rfind("blocks.", 0) == 0) {
    const size_t dot = key.find('.', 7);
    if (dot == std::string::npos) return fail(MRA_ENAME, "unknown parameter name: " + key);
    const int li = atoi(key.substr(7, dot - 7).c_str());
    if (li < 0 || li >= c.depth) return fail(MRA_ENAME, "layer index out of range: " + key);
    VitLayer& L = h->layers[li];
    const std::string sub = key.substr(dot + 1);
    if (sub == "norm1.weight") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, L.n1g, D, st); }
    else if (sub == "norm1.bias") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, L.n1b, D, st); }
    else if (sub == "norm2.weight") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, L.n2g, D, st); }
    else if (sub == "norm2.bias") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, L.n2b, D, st); }
    else if (sub == "attn.qkv.weight") { if ((rc = expect(3 * D * D))) return rc; rc = pack(L.wqkv, h->nqkv, (int)D, (int)D, 1); }
    else if (sub == "attn.q_bias" || sub == "attn.v_bias") {
      if ((rc = expect(D))) return rc;
      // [heads][hd] -> the q (or v) third of the padded bias [3][heads][96], as f32: one padded row-set of width 1
      float* dst = L.bqkv + (sub == "attn.q_bias" ? 0 : 2) * c.heads * HD_PAD;
      for (int hh = 0; hh < c.heads && !rc; ++hh)
        rc = launch_convert((const char*)src + (size_t)hh * hd * (dtype == MRA_F32 ? 4 : 2), dtype, dst + hh * HD_PAD, MRA_F32, hd, st);
    }
    else if (sub == "attn.proj.weight") { if ((rc = expect(D * D))) return rc; rc = pack(L.wproj, D, (int)D, (int)D, 0); }
    else if (sub == "attn.proj.bias") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, L.bproj, D, st); }
    else if (sub == "fc1.weight") { if ((rc = expect(I * D))) return rc; rc = pack(L.wfc1, I, (int)D, (int)D, 0); }
    else if (sub == "fc1.bias") { if ((rc = expect(I))) return rc; rc = copy_f32(src, dtype, L.bfc1, I, st); }
    else if (sub == "fc2.weight") { if ((rc = expect(D * I))) return rc; rc = pack(L.wfc2, D, (int)I, (int)I, 0); }
    else if (sub == "fc2.bias") { if ((rc = expect(D))) return rc; rc = copy_f32(src, dtype, L.bfc2, D, st); }
    else return fail(MRA_ENAME, "unknown parameter name: " + key);
  } else return fail(MRA_ENAME, "unknown parameter name: " + key);
  if (rc) return chk(rc, "vit load");
  h->loaded[key] = 1;
  h->fold_ready = false;
  return MRA_OK;
}

int mra_vit_set_option(mra_vit* h, const char* name, int32_t value) {
  if (!h || !name) return fail(MRA_EINVAL, "null argument");
  const std::string key(name);
  if (key == "ln_fold") {
    if (value != 0 && value != 1) return fail(MRA_EINVAL, "ln_fold: 0 or 1");
    h->ln_fold = value;
    return MRA_OK;
  }
  if (key == "gemm_persist") {
    if (value < 0 || value > 2) return fail(MRA_EINVAL, "gemm_persist: 0, 1 or 2 (automatic)");
    h->gemm_persist = value;
    return MRA_OK;
  }
  if (key == "attn_persist") {
    if (value != 0 && value != 1) return fail(MRA_EINVAL, "attn_persist: 0 or 1");
    h->attn_persist = value;
    return MRA_OK;
  }
  return fail(MRA_ENAME, "unknown option: " + key);
}

int mra_vit_missing(mra_vit* h) {
  if (!h) return -1;
  return 4 + 13 * h->cfg.depth - (int)h->loaded.size();
}

namespace {
// the wide buffer: patches, then Q|K|V, then the fc1 activation; with the residual stream in the operand dtype the fp32 embeddings
// are staged behind the patches inside it, so it must also hold patches + M x dim x 4 (for small patches / large images that exceeds M x wide x 2)
size_t vit_big_bytes(const mra_vit* h, size_t frames) {
  const size_t M = frames * h->S;
  const size_t wide = std::max<size_t>(std::max<size_t>(h->nqkv, h->cfg.mlp), h->kpad);
  size_t big = M * wide * 2;
  if (h->cfg.residual_dtype != MRA_F32) big = std::max(big, align_up(frames * h->np * h->np * h->kpad * 2) + M * h->cfg.dim * 4);
  return big;
}
}  // namespace

size_t mra_vit_workspace_bytes(mra_vit* h, int32_t frames) {
  if (!h || frames <= 0) return 0;
  const size_t M = (size_t)frames * h->S;
  const size_t big = vit_big_bytes(h, (size_t)frames);
  // folded LayerNorms: the op-dtype copy of the residual rows, the 128-column group statistics and (mean, rstd) per row
  // (with the stream in the operand dtype the stream is its own copy and the groups are 64 columns wide: the copy's space holds them easily)
  const size_t fold = h->can_fold() ? align_up(M * h->cfg.dim * 2) + align_up(M * (h->cfg.dim / 64) * 8) + align_up(M * 8) : 0;
  return align_up(M * h->cfg.dim * 2) + align_up(big) + fold + 4096;
}

int mra_vit_forward(mra_vit* h, const void* frames, int32_t dtype, int32_t n, void* out_, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (n < 0) return fail(MRA_EINVAL, "negative frame count");
  if (n == 0) return MRA_OK;
  if (!frames || !out_ || !workspace) return fail(MRA_EINVAL, "null argument");
  if (dtype != MRA_F32 && dtype != MRA_F16) return fail(MRA_EINVAL, "frames must be f32 or f16");
  if (mra_vit_missing(h) > 0) return fail(MRA_ESTATE, std::to_string(mra_vit_missing(h)) + " ViT parameters not loaded");
  if (workspace_bytes < mra_vit_workspace_bytes(h, n)) return fail(MRA_ENOMEM, "workspace too small: need " + std::to_string(mra_vit_workspace_bytes(h, n)));
  if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MRA_EINVAL, "workspace must be 256-byte aligned");
  const mra_vit_cfg& c = h->cfg;
  const int D = c.dim, I = c.mlp, S = h->S, op = h->op(), hd = D / c.heads;
  const long long M = (long long)n * S;
  if (M > 0x7fffffffLL / 64) return fail(MRA_EINVAL, "too many frames for one call: chunk them");
  hipStream_t st = as_stream(stream_);
  char* a16 = (char*)workspace;
  char* big = a16 + align_up((size_t)M * D * 2);
  const bool r16 = c.residual_dtype != MRA_F32;      // residual stream in the operand dtype (the reference's precision="fp16")
  float* out = r16 ? reinterpret_cast<float*>(big + align_up((size_t)n * h->np * h->np * h->kpad * 2)) : (float*)out_;   // r16: fp32 embeddings are staged, then converted
  int rc;
  {   // patches -> x[:, 1:, :] = patch GEMM + bias + pos[1:]; x[:, 0, :] = cls + pos[0]
    const long long total = (long long)n * h->np * h->np * h->kpad;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dtype == MRA_F32) {
      if (op == OP_F16) hipLaunchKernelGGL((vit_im2col_kernel<float, f16>), grid, block, 0, st, (const float*)frames, (f16*)big, total, c.img, c.patch, h->np, h->kpad);
      else hipLaunchKernelGGL((vit_im2col_kernel<float, bf16>), grid, block, 0, st, (const float*)frames, (bf16*)big, total, c.img, c.patch, h->np, h->kpad);
    } else {
      if (op == OP_F16) hipLaunchKernelGGL((vit_im2col_kernel<f16, f16>), grid, block, 0, st, (const f16*)frames, (f16*)big, total, c.img, c.patch, h->np, h->kpad);
      else hipLaunchKernelGGL((vit_im2col_kernel<f16, bf16>), grid, block, 0, st, (const f16*)frames, (bf16*)big, total, c.img, c.patch, h->np, h->kpad);
    }
    GemmProb p{};
    const int P2 = h->np * h->np;
    p.A = big; p.a = plain(n * P2, h->kpad);
    p.W = h->wpatch; p.bias = h->bpatch;
    p.R = h->pos + D; p.r = items_view(0, P2, D);
    p.C = out + D; p.c = items_view((long long)S * D, P2, D);
    p.M = n * P2; p.N = D; p.K = h->kpad;
    rc = launch_gemm(&p, 1, EPI_RES_F32, op, st);
    if (rc) return chk(rc, "patch embedding gemm");
    hipLaunchKernelGGL(vit_cls_kernel, dim3((unsigned)(((long long)n * D + 255) / 256)), dim3(256), 0, st, out, h->cls, h->pos, n, S, D);
    if (r16 && (rc = launch_convert(out, MRA_F32, out_, c.op_dtype, M * D, st))) return chk(rc, "embedding convert");
  }
  const int xdt = r16 ? c.op_dtype : MRA_F32;       // dtype code of the residual stream for the LayerNorm kernel
  void* x = r16 ? out_ : (void*)out;
  const size_t attn_lds = 2 * KS_PAD * KV_PITCH;
  static unsigned long long attr_done = 0;
  if (!(attr_done >> (h->device & 63) & 1)) {
    if (hipFuncSetAttribute((const void*)vit_attn_kernel<f16, 257>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vit_attn_kernel<bf16, 257>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vit_attn_kernel<f16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vit_attn_kernel<bf16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vit_attn_persist_kernel<f16, 257>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vit_attn_persist_kernel<bf16, 257>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds) != hipSuccess)
      return fail(MRA_EHIP, "hipFuncSetAttribute(vit_attn_kernel)");
    attr_done |= 1ull << (h->device & 63);
  }
  const float sl2 = LOG2E / sqrtf((float)hd);
  // folded LayerNorms (see vit_fold_weight_kernel): x16 = the op-dtype copy of the residual rows, groups = their 128-column statistics as the
  // residual GEMMs leave them, rstat = (mean, rstd) per row for the QKV / fc1 epilogues
  const bool fold = h->ln_fold && h->can_fold();
  char* x16 = big + align_up(vit_big_bytes(h, (size_t)n));
  float2* groups = reinterpret_cast<float2*>(x16 + align_up((size_t)M * D * 2));
  float2* rstat = reinterpret_cast<float2*>(reinterpret_cast<char*>(groups) + align_up((size_t)M * (D / 64) * 8));
  const int gsz = r16 ? 64 : 128;   // columns per statistics group: a lane octet of the staged copy-out / a wave's half of the tile
  if (fold) {
    if (!h->fold_ready && (rc = fold_weights(h, st))) return chk(rc, "vit LayerNorm fold of the weights");
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    if (r16) {
      x16 = (char*)x;   // the stream is the operand
      if (op == OP_F16) hipLaunchKernelGGL(vit_row_stats16_kernel<f16>, grid, block, 0, st, (const f16*)x, M, D, c.ln_eps, rstat);
      else hipLaunchKernelGGL(vit_row_stats16_kernel<bf16>, grid, block, 0, st, (const bf16*)x, M, D, c.ln_eps, rstat);
    } else {
      if (op == OP_F16) hipLaunchKernelGGL(vit_row_stats_kernel<f16>, grid, block, 0, st, out, M, D, c.ln_eps, rstat, (f16*)x16);
      else hipLaunchKernelGGL(vit_row_stats_kernel<bf16>, grid, block, 0, st, out, M, D, c.ln_eps, rstat, (bf16*)x16);
    }
  }
  auto persist = [&](long long tiles) { return h->gemm_persist == 1 || (h->gemm_persist == 2 && h->cus > 0 && tiles <= 64LL * h->cus) ? 1 : 0; };
  auto row_stats = [&]() {
    hipLaunchKernelGGL(vit_group_stats_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, groups, M, D / gsz, (float)gsz, c.ln_eps, rstat);
  };
  for (int li = 0; li < c.depth; ++li) {
    const VitLayer& L = h->layers[li];
    if (!fold) {
      rc = launch_modality_ln(x, xdt, nullptr, n, S, D, L.n1g, L.n1b, c.ln_eps, a16, op, st);
      if (rc) return chk(rc, "vit ln1");
    }
    {
      GemmProb p{};
      p.A = fold ? x16 : a16; p.a = plain((int)M, D); p.W = fold ? L.wqkv_f : L.wqkv; p.bias = fold ? L.bf_qkv : L.bqkv;
      p.C = big; p.c = plain((int)M, h->nqkv); p.M = (int)M; p.N = h->nqkv; p.K = D;
      if (fold) { p.ln_gain = L.cs_qkv; p.ln_y32 = reinterpret_cast<float*>(rstat); p.tile_cfg = 3; }
      p.persist = persist((M + 255) / 256 * (h->nqkv / 256));
      rc = launch_gemm(&p, 1, fold ? EPI_LNF_OP : EPI_OP, op, st);
      if (rc) return chk(rc, "vit qkv gemm");
    }
    if (S == 257 && h->attn_persist && h->cus > 0) {   // one persistent workgroup per CU, the next unit's K / V prefetched
      const int units = n * c.heads, grid = std::min(units, h->cus);
      if (op == OP_F16) hipLaunchKernelGGL((vit_attn_persist_kernel<f16, 257>), dim3(grid), dim3(ATTP_WAVES * 64), attn_lds, st, (const f16*)big, (f16*)a16, units, c.heads, hd, sl2);
      else hipLaunchKernelGGL((vit_attn_persist_kernel<bf16, 257>), dim3(grid), dim3(ATTP_WAVES * 64), attn_lds, st, (const bf16*)big, (bf16*)a16, units, c.heads, hd, sl2);
    } else if (S == 257) {   // ViT-g/224: the sequence length as a compile-time constant
      if (op == OP_F16) hipLaunchKernelGGL((vit_attn_kernel<f16, 257>), dim3(n * c.heads), dim3(ATT_WAVES * 64), attn_lds, st, (const f16*)big, (f16*)a16, S, c.heads, hd, sl2);
      else hipLaunchKernelGGL((vit_attn_kernel<bf16, 257>), dim3(n * c.heads), dim3(ATT_WAVES * 64), attn_lds, st, (const bf16*)big, (bf16*)a16, S, c.heads, hd, sl2);
    } else {
      if (op == OP_F16) hipLaunchKernelGGL((vit_attn_kernel<f16, 0>), dim3(n * c.heads), dim3(ATT_WAVES * 64), attn_lds, st, (const f16*)big, (f16*)a16, S, c.heads, hd, sl2);
      else hipLaunchKernelGGL((vit_attn_kernel<bf16, 0>), dim3(n * c.heads), dim3(ATT_WAVES * 64), attn_lds, st, (const bf16*)big, (bf16*)a16, S, c.heads, hd, sl2);
    }
    // x += A W^T + b for the two N = dim GEMMs.  dim = 1408 is 5.5 tiles of 256: GemmProb::tile_cfg 8 runs the five full column tiles
    // of two row tiles and then their last 128 columns as one 128 x 512 tile, all in one launch (a masked sixth 256-wide tile wastes 9 %)
    auto residual_gemm = [&](const void* A, int K, const void* W, const float* bias, bool stat) {
      GemmProb p{};
      p.A = A; p.a = plain((int)M, K); p.W = W; p.bias = bias;
      p.R = out; p.r = plain((int)M, D); p.C = x; p.c = plain((int)M, D); p.aux = x;
      p.M = (int)M; p.N = D; p.K = K; p.tile_cfg = h->proj_tile; p.n_mask = h->proj_tile == 3; p.order = 8;
      int epi = r16 ? EPI_RES_OP : EPI_RES_F32;
      const bool split = h->proj_tile == 3 && h->tail_tile && D % 256 == 128 && D > 128 && K % 128 == 0;
      if (split) { p.tile_cfg = 8; p.n_mask = 0; p.order = 0; }
      if (stat) {   // the rows' op-dtype copy and group statistics for the folded LayerNorm behind this GEMM (can_fold() implies split)
        p.ln_y32 = reinterpret_cast<float*>(groups);
        if (r16) epi = EPI_RES_OP_STAT;
        else { p.ln_y16 = x16; p.ln_y16v = plain((int)M, D); epi = EPI_RES_F32_STAT; }
      }
      return launch_gemm(&p, 1, epi, op, st);
    };
    rc = residual_gemm(a16, D, L.wproj, L.bproj, fold);
    if (rc) return chk(rc, "vit projection gemm");
    if (fold) {
      row_stats();
    } else {
      rc = launch_modality_ln(x, xdt, nullptr, n, S, D, L.n2g, L.n2b, c.ln_eps, a16, op, st);
      if (rc) return chk(rc, "vit ln2");
    }
    {
      GemmProb p{};
      p.A = fold ? x16 : a16; p.a = plain((int)M, D); p.W = fold ? L.wfc1_f : L.wfc1; p.bias = fold ? L.bf_fc1 : L.bfc1;
      p.C = big; p.c = plain((int)M, I); p.M = (int)M; p.N = I; p.K = D;
      if (fold) { p.ln_gain = L.cs_fc1; p.ln_y32 = reinterpret_cast<float*>(rstat); p.tile_cfg = 3; }
      p.persist = persist((M + 255) / 256 * (I / 256));
      rc = launch_gemm(&p, 1, fold ? EPI_LNF_GELU_OP : EPI_GELU_OP, op, st);
      if (rc) return chk(rc, "vit fc1 gemm");
    }
    const bool feeds_ln = fold && li + 1 < c.depth;   // the last block's rows are the output: nothing reads their copy
    rc = residual_gemm(big, I, L.wfc2, L.bfc2, feeds_ln);
    if (rc) return chk(rc, "vit fc2 gemm");
    if (feeds_ln) row_stats();
  }
  return hipGetLastError() == hipSuccess ? MRA_OK : fail(MRA_EHIP, "vit forward launch");
}

double mra_vit_flops(mra_vit* h, int32_t frames) {
  if (!h) return 0.0;
  const double n = h->S, d = h->cfg.dim, I = h->cfg.mlp;
  const double per_block = 2 * n * d * 3 * d + 4 * n * n * d + 2 * n * d * d + 4 * n * d * I;
  return frames * (h->cfg.depth * per_block + 2.0 * (n - 1) * 3 * h->cfg.patch * h->cfg.patch * d);
}

}  // extern "C"
