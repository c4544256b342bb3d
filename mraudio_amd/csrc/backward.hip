// Backward-pass kernels of the Q-Former (BASELINE config 5: "Q-Former fwd+bwd with fused LN/GELU grads").
// The reference never back-propagates through its Q-Formers (they are frozen,
// models/xinstructblip.py:196-204); the definition of "right" here is torch.autograd over the CPU oracle
// (tests/test_gpu_backward.py).  Sizes are a fine-tuning batch (items * S ~ 1 k rows), so these kernels
// favour simplicity; the two weight-gradient / data-gradient GEMMs (gemm_tn.hip, gemm.hip) carry the flops.
//
//   ln_bwd_kernel        dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma, statistics
//                        recomputed from the saved pre-LN row; dgamma / dbeta by per-workgroup partial sums
//                        in registers + LDS, then one float atomic per column and workgroup.
//   gelu_fwd / gelu_bwd  exact (erf) GELU and its derivative, elementwise on the up-projection.
//   attn_bwd_kernel      flash-style recompute from (Q, K, V, O, dO, LSE): one workgroup per (item, head),
//                        KV tiles outer / 32-row query blocks inner, dK/dV of a tile and dQ of the whole
//                        head accumulate in LDS (fp32): no atomics, reproducible.  fp32 VALU math.
//   embed_bwd_kernel     gradients of query tokens, position and word embeddings (row atomics for words).
//   transpose16_kernel   W [R][C] -> W^T [C][R] (operand dtype) for the data-gradient GEMMs.
#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

__device__ __forceinline__ long long brow(const RowView& v, int m) {
  const int item = m / v.rpi;
  return (long long)item * v.item_stride + (long long)(m - item * v.rpi) * v.ld;
}

template <typename T>
__device__ __forceinline__ void st4(T* p, f32x4 v) {
  typename Vec4<T>::type o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
  *reinterpret_cast<typename Vec4<T>::type*>(p) = o;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  H = 256 * NV.  RPW rows per wave, 4 waves per workgroup.
// ---------------------------------------------------------------------------------------------
// Two independent jobs in one launch (the query-row and the text-row LayerNorm of a Q-Former layer): workgroups [0, wg_a) take `a0`, the
// rest `a1`.  A single job: wg_a = gridDim.x.
template <typename T, int NV>
__global__ void __launch_bounds__(256) ln_bwd_kernel(LnBwdArgs a0, LnBwdArgs a1, int wg_a) {
  constexpr int H = NV * 256, RPW = 4;
  __shared__ float red[2][4][H];
  const bool second = (int)blockIdx.x >= wg_a;          // workgroup-uniform
  const LnBwdArgs& a = second ? a1 : a0;
  const int wg = second ? blockIdx.x - wg_a : blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = dg[i]; }
  f32x4 gam[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) gam[i] = *reinterpret_cast<const f32x4*>(a.gamma + (i * 64 + lane) * 4);
  // all RPW rows of this wave are fetched up front (rows past the end re-read the last row and are skipped below): one memory round trip
  // per wave instead of one per row -- a launch over ~1 k rows is a single round of latency-bound workgroups
  const int row0 = (wg * 4 + wave) * RPW;
  f32x4 xs[RPW][NV], dys[RPW][NV];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = min(row0 + rr, a.rows - 1);
    const float* xr = a.x + brow(a.xv, row);
    const float* dyr = a.dy + brow(a.dyv, row);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xs[rr][i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
      dys[rr][i] = *reinterpret_cast<const f32x4*>(dyr + (i * 64 + lane) * 4);
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr;
    if (row >= a.rows) break;
    f32x4 x[NV], dy[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[i] = xs[rr][i];
      dy[i] = dys[rr][i];
      s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    }
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[i][e] -= mean; q += x[i][e] * x[i][e]; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + a.eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[i][e] *= rstd;  // xhat
        const float g = dy[i][e] * gam[i][e];
        sg += g;
        sgx += g * x[i][e];
        dg[i][e] += dy[i][e] * x[i][e];
        db[i][e] += dy[i][e];
      }
    sg = wave_sum(sg) * (1.0f / H);
    sgx = wave_sum(sgx) * (1.0f / H);
    float* dxr = a.dx + brow(a.dxv, row);
    const float* addr = a.add ? a.add + brow(a.addv, row) : nullptr;
    T* dx16 = a.dx16 ? (T*)a.dx16 + brow(a.dx16v, row) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = rstd * (dy[i][e] * gam[i][e] - sg - x[i][e] * sgx);
      if (addr) d += *reinterpret_cast<const f32x4*>(addr + c);
      *reinterpret_cast<f32x4*>(dxr + c) = d;
      if (dx16) st4<T>(dx16 + c, d);
    }
  }
  if (!a.dgamma) return;
  // workgroup reduction of the parameter gradients, then one atomic per column
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    *reinterpret_cast<f32x4*>(&red[0][wave][(i * 64 + lane) * 4]) = dg[i];
    *reinterpret_cast<f32x4*>(&red[1][wave][(i * 64 + lane) * 4]) = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(a.dgamma + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
    atomicAdd(a.dbeta + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
  }
}

// ---------------------------------------------------------------------------------------------
// GELU
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_grad(float u) { return gelu_erf_grad(u); }

template <typename T, bool BWD>
__global__ void __launch_bounds__(256) gelu_kernel(const T* u, const T* df, T* out, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const typename Vec8<T>::type uv = *reinterpret_cast<const typename Vec8<T>::type*>(u + i * 8);
    typename Vec8<T>::type o;
    if (BWD) {
      const typename Vec8<T>::type dv = *reinterpret_cast<const typename Vec8<T>::type*>(df + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>((float)dv[e] * gelu_grad((float)uv[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(gelu_erf((float)uv[e]));
    }
    *reinterpret_cast<typename Vec8<T>::type*>(out + i * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// attention backward
// ---------------------------------------------------------------------------------------------
constexpr int QB = 32, KB = 32, DH = 64, PADW = 65;  // block sizes; LDS rows padded to 65 floats
constexpr float LOG2E_B = 1.4426950408889634f;

template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x;
  const int head = blockIdx.x % a.heads, item = blockIdx.x / a.heads;
  const int nqb = (a.q_rows + QB - 1) / QB, nkb = (a.kv_len + KB - 1) / KB;
  float* dQs = smem;                       // [nqb*QB][PADW]
  float* Qs = dQs + nqb * QB * PADW;       // [QB][PADW]
  float* dOs = Qs + QB * PADW;
  float* Ks = dOs + QB * PADW;             // [KB][PADW]
  float* Vs = Ks + KB * PADW;
  float* dKs = Vs + KB * PADW;
  float* dVs = dKs + KB * PADW;
  float* Ps = dVs + KB * PADW;             // [QB][KB + 1]
  float* dSs = Ps + QB * (KB + 1);
  float* delta = dSs + QB * (KB + 1);      // [nqb*QB]
  float* lse = delta + nqb * QB;           // [nqb*QB]

  const T* Qg = (const T*)a.Q + (long long)item * a.q_item_stride + head * DH;
  const T* dOg = (const T*)a.dO + (long long)item * a.o_item_stride + head * DH;
  const T* Og = (const T*)a.O + (long long)item * a.o_item_stride + head * DH;
  const T* Kg = (const T*)a.K + (long long)item * a.k_item_stride + (long long)head * a.k_head_stride;
  const T* Vg = (const T*)a.V + (long long)item * a.v_item_stride + (long long)head * a.v_head_stride;
  T* dQg = (T*)a.dQ + (long long)item * a.dq_item_stride + head * DH;
  T* dKg = (T*)a.dK + (long long)item * a.dk_item_stride + (long long)head * a.dk_head_stride;
  T* dVg = (T*)a.dV + (long long)item * a.dv_item_stride + (long long)head * a.dv_head_stride;

  for (int i = tid; i < nqb * QB * PADW; i += 256) dQs[i] = 0.f;
  // delta[q] = sum_d dO[q][d] * O[q][d]; lse from the forward (log2 units, mask included)
  for (int q = tid; q < nqb * QB; q += 256) {
    float s = 0.f;
    if (q < a.q_rows) {
      for (int d = 0; d < DH; ++d) s += (float)dOg[(long long)q * a.o_ld + d] * (float)Og[(long long)q * a.o_ld + d];
      lse[q] = a.lse[((long long)item * a.heads + head) * a.q_rows + q];
    } else {
      lse[q] = 0.f;
    }
    delta[q] = s;
  }
  __syncthreads();
  const float sl2 = a.scale * LOG2E_B;
  for (int kb = 0; kb < nkb; ++kb) {
    // K / V tile (rows past kv_len zero), dK / dV accumulators
    for (int i = tid; i < KB * DH; i += 256) {
      const int t = i >> 6, d = i & 63, tok = kb * KB + t;
      float kv = 0.f, vv = 0.f;
      if (tok < a.kv_len) { kv = (float)Kg[(long long)tok * a.k_ld + d]; vv = (float)Vg[(long long)tok * a.v_ld + d]; }
      Ks[t * PADW + d] = kv; Vs[t * PADW + d] = vv;
      dKs[t * PADW + d] = 0.f; dVs[t * PADW + d] = 0.f;
    }
    for (int qb = 0; qb < nqb; ++qb) {
      __syncthreads();
      for (int i = tid; i < QB * DH; i += 256) {
        const int q = i >> 6, d = i & 63, qr = qb * QB + q;
        float qv = 0.f, dv = 0.f;
        if (qr < a.q_rows) { qv = (float)Qg[(long long)qr * a.q_ld + d]; dv = (float)dOg[(long long)qr * a.o_ld + d]; }
        Qs[q * PADW + d] = qv; dOs[q * PADW + d] = dv;
      }
      __syncthreads();
      // P and dS: 1024 (q, t) pairs, 4 per thread
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = tid + r * 256, q = idx >> 5, t = idx & 31;
        const int qr = qb * QB + q, tok = kb * KB + t;
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < DH; ++d) {
          s += Qs[q * PADW + d] * Ks[t * PADW + d];
          dp += dOs[q * PADW + d] * Vs[t * PADW + d];
        }
        float p = 0.f;
        if (qr < a.q_rows && tok < a.kv_len) {
          float y = s * sl2;
          if (a.mask) y += (1.0f - (float)a.mask[(long long)item * a.mask_ld + tok]) * (-10000.0f * LOG2E_B);
          p = __builtin_amdgcn_exp2f(y - lse[qr]);
        }
        Ps[q * (KB + 1) + t] = p;
        dSs[q * (KB + 1) + t] = p * (dp - delta[qr]) * a.scale;
      }
      __syncthreads();
      // dV[t][d] += sum_q P[q][t] dO[q][d];  dK[t][d] += sum_q dS[q][t] Q[q][d]   (2048 outputs, 8 per thread)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int idx = tid + r * 256, t = idx >> 6, d = idx & 63;
        float av = 0.f, ak = 0.f;
        for (int q = 0; q < QB; ++q) {
          av += Ps[q * (KB + 1) + t] * dOs[q * PADW + d];
          ak += dSs[q * (KB + 1) + t] * Qs[q * PADW + d];
        }
        dVs[t * PADW + d] += av;
        dKs[t * PADW + d] += ak;
      }
      // dQ[q][d] += sum_t dS[q][t] K[t][d]
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int idx = tid + r * 256, q = idx >> 6, d = idx & 63;
        float aq = 0.f;
        for (int t = 0; t < KB; ++t) aq += dSs[q * (KB + 1) + t] * Ks[t * PADW + d];
        dQs[(qb * QB + q) * PADW + d] += aq;
      }
    }
    __syncthreads();
    for (int i = tid; i < KB * DH; i += 256) {
      const int t = i >> 6, d = i & 63, tok = kb * KB + t;
      if (tok < a.kv_len) {
        dKg[(long long)tok * a.dk_ld + d] = from_f32<T>(dKs[t * PADW + d]);
        dVg[(long long)tok * a.dv_ld + d] = from_f32<T>(dVs[t * PADW + d]);
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < nqb * QB * DH; i += 256) {
    const int q = i >> 6, d = i & 63;
    if (q < a.q_rows) dQg[(long long)q * a.dq_ld + d] = from_f32<T>(dQs[q * PADW + d]);
  }
}

// ---------------------------------------------------------------------------------------------
// attention backward on the matrix cores (v_mfma_f32_32x32x16, fp32 accumulate)
// ---------------------------------------------------------------------------------------------
// One workgroup per (item, head), 4 waves.  A wave owns one 32-token K/V block per round (4 blocks in
// flight), keeps dK / dV of that block in registers and walks the 32-row query blocks; the Q and dO tiles
// of the current query block are shared.  All tiles are [32 rows][64 d] 16-bit, 128-byte rows, staged by
// LDS-DMA with the 16-byte-chunk swizzle c ^ (((row >> 1) & 1) << 2) on the source address (the V tile
// of attention.hip / the operand tiles of gemm_tn.hip).  Two kinds of fragment come out of a tile:
//   direct  F_X[ks]   lane (row l & 31) reads 16 bytes = d 16 ks + 8 (l >> 5) .. + 7     (contraction over d)
//   tr      T_X[ct,s] ds_read_b64_tr_b16: lane (column 32 ct + (l & 31)) gets rows 16 s + {4h..4h+3, 8+4h..8+4h+3}
//                     -- exactly the row order of an MFMA accumulator's registers, so P / dS go from the
//                     accumulator to the next MFMA's A operand without touching LDS   (contraction over rows)
// Both orientations of the score tile are computed (S = Q K^T with the token on the lane axis feeds dV and
// dK, S^T = K Q^T with the query on the lane axis feeds dQ): 8 extra MFMAs per tile pair instead of a
// transpose through LDS.  dQ partials of the four waves meet in an fp32 LDS tile through ds_add_f32.
namespace bwd {
constexpr int TILE = 32 * 128;  // bytes of one [32][64] 16-bit tile

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type direct_frag(const char* tile, int lane, int ks) {
  const int r = lane & 31, c = 2 * ks + (lane >> 5);
  return lds_read8<T>(tile + r * 128 + ((c ^ (((r >> 1) & 1) << 2)) << 4));
}

// both k-steps (rows 0-15, 16-31) of column tile ct, as two 8-element operands
template <typename T>
__device__ __forceinline__ void tr_frags(unsigned tile, unsigned lane_off_ct, typename Vec8<T>::type& s0, typename Vec8<T>::type& s1) {
  i16x4 a0, a1, a2, a3;
  const unsigned addr = tile + lane_off_ct;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4\n\t"
      "ds_read_b64_tr_b16 %1, %4 offset:1024\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:2048\n\t"
      "ds_read_b64_tr_b16 %3, %4 offset:3072\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
      : "v"(addr)
      : "memory");
  i16x8 v0, v1;
  v0[0] = a0[0]; v0[1] = a0[1]; v0[2] = a0[2]; v0[3] = a0[3]; v0[4] = a1[0]; v0[5] = a1[1]; v0[6] = a1[2]; v0[7] = a1[3];
  v1[0] = a2[0]; v1[1] = a2[1]; v1[2] = a2[2]; v1[3] = a2[3]; v1[4] = a3[0]; v1[5] = a3[1]; v1[6] = a3[2]; v1[7] = a3[3];
  s0 = __builtin_bit_cast(typename Vec8<T>::type, v0);
  s1 = __builtin_bit_cast(typename Vec8<T>::type, v1);
}
}  // namespace bwd

template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_mfma_kernel(const AttnBwdArgs a) {
  using V8 = typename Vec8<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int head = blockIdx.x % a.heads, item = blockIdx.x / a.heads;
  const int nqb = (a.q_rows + 31) / 32, nkb = (a.kv_len + 31) / 32;
  const int h = lane >> 5, ln = lane & 31;

  char* Qs = smem_raw;                                   // shared Q tile
  char* dOs = Qs + bwd::TILE;                            // shared dO tile
  char* KVs = dOs + bwd::TILE;                           // per wave: K tile, V tile
  float* dQs = reinterpret_cast<float*>(KVs + 4 * 2 * bwd::TILE);   // [nqb*32][64]
  float* lse_s = dQs + nqb * 32 * 64;                    // [nqb*32]
  float* delta_s = lse_s + nqb * 32;                     // [nqb*32]
  float* mterm_s = delta_s + nqb * 32;                   // [4][32]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
  const unsigned q_tile = lds0, do_tile = lds0 + bwd::TILE;
  const unsigned k_tile = lds0 + 2 * bwd::TILE + wave * 2 * bwd::TILE, v_tile = k_tile + bwd::TILE;
  const char* Kw = KVs + wave * 2 * bwd::TILE;   // this wave's K tile, V tile behind it
  const char* Vw = Kw + bwd::TILE;

  const T* Qg = (const T*)a.Q + (long long)item * a.q_item_stride + head * 64;
  const T* dOg = (const T*)a.dO + (long long)item * a.o_item_stride + head * 64;
  const T* Og = (const T*)a.O + (long long)item * a.o_item_stride + head * 64;
  const T* Kg = (const T*)a.K + (long long)item * a.k_item_stride + (long long)head * a.k_head_stride;
  const T* Vg = (const T*)a.V + (long long)item * a.v_item_stride + (long long)head * a.v_head_stride;
  T* dQg = (T*)a.dQ + (long long)item * a.dq_item_stride + head * 64;
  T* dKg = (T*)a.dK + (long long)item * a.dk_item_stride + (long long)head * a.dk_head_stride;
  T* dVg = (T*)a.dV + (long long)item * a.dv_item_stride + (long long)head * a.dv_head_stride;

  for (int i = tid; i < nqb * 32 * 64; i += 256) dQs[i] = 0.f;
  // delta[q] = sum_d dO[q][d] O[q][d]: 8 lanes per row, 16 bytes each
  for (int base = 0; base < nqb * 32; base += 32) {
    const int q = base + (tid >> 3), c = tid & 7;
    float sacc = 0.f;
    if (q < a.q_rows) {
      const V8 o = *reinterpret_cast<const V8*>(Og + (long long)q * a.o_ld + c * 8);
      const V8 g = *reinterpret_cast<const V8*>(dOg + (long long)q * a.o_ld + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) sacc += (float)o[j] * (float)g[j];
    }
    sacc += __shfl_xor(sacc, 1);
    sacc += __shfl_xor(sacc, 2);
    sacc += __shfl_xor(sacc, 4);
    if (c == 0) {
      delta_s[q] = sacc;
      lse_s[q] = q < a.q_rows ? a.lse[((long long)item * a.heads + head) * a.q_rows + q] : 0.f;
    }
  }
  // transposed-read lane offsets for the two 32-column halves of a tile (gemm_tn.hip)
  unsigned tr_off[2];
  {
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, p = i & 3;
    const int row = 4 * h + q4;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int c = 4 * ct + 2 * (g & 1) + (p >> 1);
      tr_off[ct] = row * 128 + ((c ^ (((q4 >> 1) & 1) << 2)) << 4) + (p & 1) * 8;
    }
  }
  const float sl2 = a.scale * LOG2E_B;
  const int nrounds = (nkb + 3) / 4;
  for (int rnd = 0; rnd < nrounds; ++rnd) {
    const int kb = rnd * 4 + wave;
    const bool active = kb < nkb;   // wave-uniform
    const int t0 = kb * 32;
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), sc = lane & 7;
        const int tok = min(t0 + row, a.kv_len - 1);
        const int src = (sc ^ (((row >> 1) & 1) << 2)) * 8;
        glds16(Kg + (long long)tok * a.k_ld + src, (char*)Kw + i * 1024);
        glds16(Vg + (long long)tok * a.v_ld + src, (char*)Vw + i * 1024);
      }
      if (lane < 32) {
        const int tok = t0 + lane;
        float m = 0.f;
        if (a.mask && tok < a.kv_len) m = (1.0f - (float)a.mask[(long long)item * a.mask_ld + tok]) * (-10000.0f * LOG2E_B);
        mterm_s[wave * 32 + lane] = m;
      }
    }
    f32x16 accK[2], accV[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { accK[0][i] = 0.f; accK[1][i] = 0.f; accV[0][i] = 0.f; accV[1][i] = 0.f; }
    for (int qb = 0; qb < nqb; ++qb) {
      __syncthreads();   // everyone is done with the previous Q / dO tiles
      {
        const int row = tid >> 3, sc = tid & 7;
        const int q = min(qb * 32 + row, a.q_rows - 1);
        const int src = (sc ^ (((row >> 1) & 1) << 2)) * 8;
        glds16(Qg + (long long)q * a.q_ld + src, Qs + wave * 1024);
        glds16(dOg + (long long)q * a.o_ld + src, dOs + wave * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (!active) continue;
      const int q0 = qb * 32;
      // ---- token on the lane axis: S[q][t], dP[q][t] -> P, dS -> dV, dK ------------------------------
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32<T>(bwd::direct_frag<T>(Qs, lane, ks), bwd::direct_frag<T>(Kw, lane, ks), s);
        dp = mfma32<T>(bwd::direct_frag<T>(dOs, lane, ks), bwd::direct_frag<T>(Vw, lane, ks), dp);
      }
      {
        const float mt = mterm_s[wave * 32 + ln];
        const bool tok_ok = t0 + ln < a.kv_len;
        V8 p8[2], ds8[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          float p = 0.f;
          if (tok_ok && q < a.q_rows) p = __builtin_amdgcn_exp2f(s[r] * sl2 + mt - lse_s[q]);
          const float d = p * (dp[r] - delta_s[q]) * a.scale;
          p8[r >> 3][r & 7] = from_f32<T>(p);
          ds8[r >> 3][r & 7] = from_f32<T>(d);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          V8 b0, b1;
          bwd::tr_frags<T>(do_tile, tr_off[ct], b0, b1);
          accV[ct] = mfma32<T>(p8[0], b0, accV[ct]);
          accV[ct] = mfma32<T>(p8[1], b1, accV[ct]);
          bwd::tr_frags<T>(q_tile, tr_off[ct], b0, b1);
          accK[ct] = mfma32<T>(ds8[0], b0, accK[ct]);
          accK[ct] = mfma32<T>(ds8[1], b1, accK[ct]);
        }
      }
      // ---- query on the lane axis: S^T[t][q], dP^T[t][q] -> dS^T -> dQ ---------------------------------
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32<T>(bwd::direct_frag<T>(Kw, lane, ks), bwd::direct_frag<T>(Qs, lane, ks), s);
        dp = mfma32<T>(bwd::direct_frag<T>(Vw, lane, ks), bwd::direct_frag<T>(dOs, lane, ks), dp);
      }
      {
        const int q = q0 + ln;
        const bool q_ok = q < a.q_rows;
        const float lq = lse_s[q], dq = delta_s[q];
        V8 ds8[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tl = (r & 3) + 8 * (r >> 2) + 4 * h;
          float p = 0.f;
          if (q_ok && t0 + tl < a.kv_len) p = __builtin_amdgcn_exp2f(s[r] * sl2 + mterm_s[wave * 32 + tl] - lq);
          ds8[r >> 3][r & 7] = from_f32<T>(p * (dp[r] - dq) * a.scale);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          V8 b0, b1;
          bwd::tr_frags<T>(k_tile, tr_off[ct], b0, b1);
          f32x16 accQ;
#pragma unroll
          for (int i = 0; i < 16; ++i) accQ[i] = 0.f;
          accQ = mfma32<T>(ds8[0], b0, accQ);
          accQ = mfma32<T>(ds8[1], b1, accQ);
          // rows q (registers), column d = 32 ct + ln
          float* dst = dQs + (size_t)q0 * 64 + 32 * ct + ln;
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(dst + ((r & 3) + 8 * (r >> 2) + 4 * h) * 64, accQ[r]);
        }
      }
    }
    if (active) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tok = t0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (tok < a.kv_len) {
            dKg[(long long)tok * a.dk_ld + 32 * ct + ln] = from_f32<T>(accK[ct][r]);
            dVg[(long long)tok * a.dv_ld + 32 * ct + ln] = from_f32<T>(accV[ct][r]);
          }
        }
    }
  }
  __syncthreads();
  for (int i = tid; i < a.q_rows * 8; i += 256) {
    const int q = i >> 3, c = (i & 7) * 8;
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(dQs[q * 64 + c + j]);
    *reinterpret_cast<V8*>(dQg + (long long)q * a.dq_ld + c) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// embeddings backward: d_emb [items][Q + L][H] f32
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_bwd_kernel(const float* demb, const long long* ids, int items, int L, int Q, int H,
                                                        int vocab, float* dquery, float* dpos, float* dword) {
  // one workgroup per sequence position s; threads over columns; loop over items
  const int s = blockIdx.x, S = Q + L;
  for (int c = threadIdx.x; c < H; c += 256) {
    float acc = 0.f;
    for (int n = 0; n < items; ++n) {
      const float g = demb[((long long)n * S + s) * H + c];
      acc += g;
      if (s >= Q && dword) {
        long long id = ids[(long long)n * L + (s - Q)];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        atomicAdd(dword + id * H + c, g);
      }
    }
    if (s < Q) { if (dquery) dquery[(long long)s * H + c] += acc; }
    else if (dpos) dpos[(long long)(s - Q) * H + c] += acc;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) transpose16_kernel(const T* src, T* dst, int R, int C) {
  __shared__ T tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (by + j < R && bx + tx < C) tile[j][tx] = src[(long long)(by + j) * C + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < C && by + tx < R) dst[(long long)(bx + j) * R + by + tx] = tile[tx][j];
}

template <typename T>
__global__ void __launch_bounds__(256) transpose16_batch_kernel(const TrJob* jobs, int njobs) {
  __shared__ T tile[32][33];
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;                 // last job whose tile_begin <= b (workgroup-uniform binary search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile_begin <= b) lo = mid; else hi = mid - 1;
  }
  const TrJob j = jobs[lo];
  const int local = b - j.tile_begin;
  const int bx = (local % j.tiles_x) * 32, by = (local / j.tiles_x) * 32;
  const T* src = (const T*)j.src;
  T* dst = (T*)j.dst;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    if (by + r < j.R && bx + tx < j.C) tile[r][tx] = src[(long long)(by + r) * j.C + bx + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bx + r < j.C && by + tx < j.R) dst[(long long)(bx + r) * j.R + by + tx] = tile[tx][r];
}

__device__ __forceinline__ f32x4 adam_update(f32x4& p, f32x4& g, f32x4& m, f32x4& v, const AdamScalars& sc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ge = g[e] + sc.weight_decay * p[e];
    m[e] = sc.beta1 * m[e] + (1.f - sc.beta1) * ge;
    v[e] = sc.beta2 * v[e] + (1.f - sc.beta2) * ge * ge;
    p[e] -= sc.step_size * m[e] / (sqrtf(v[e]) * sc.inv_sqrt_bc2 + sc.eps);
  }
  return p;
}

// matrices with a transposed training copy: one 64 x 256 tile per workgroup (R a multiple of 64, C of 256): 1 KiB contiguous per row and array
// (64 x 64 tiles -- 256-byte runs over nine streams -- ran the pass at 3.5 TB/s), four 16-row steps with 16 loads per thread in flight
template <typename T>
__global__ void __launch_bounds__(256) adam_mat_kernel(float* master, float* grad, float* mom, float* var, const AdamMatJob* jobs, int njobs, AdamScalars sc) {
  __shared__ T tile[64][264];
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;                 // last job whose tile_begin <= b (workgroup-uniform binary search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile_begin <= b) lo = mid; else hi = mid - 1;
  }
  const AdamMatJob j = jobs[lo];
  const int local = b - j.tile_begin;
  const int bx = (local % j.tiles_x) * 256, by = (local / j.tiles_x) * 64;
  const int t = threadIdx.x, c4 = (t & 63) * 4, r0 = t >> 6;   // a wave = one 1 KiB row piece
  T* dst = (T*)j.dst;
#pragma unroll 1
  for (int step = 0; step < 4; ++step) {
    f32x4 p[4], g[4], m[4], v[4];
    long long o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = step * 16 + it * 4 + r0;
      o[it] = (long long)j.off + (long long)(by + r) * j.C + bx + c4;
      p[it] = *reinterpret_cast<const f32x4*>(master + o[it]);
      g[it] = *reinterpret_cast<const f32x4*>(grad + o[it]);
      m[it] = *reinterpret_cast<const f32x4*>(mom + o[it]);
      v[it] = *reinterpret_cast<const f32x4*>(var + o[it]);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = step * 16 + it * 4 + r0;
      adam_update(p[it], g[it], m[it], v[it], sc);
      *reinterpret_cast<f32x4*>(master + o[it]) = p[it];
      *reinterpret_cast<f32x4*>(mom + o[it]) = m[it];
      *reinterpret_cast<f32x4*>(var + o[it]) = v[it];
      if (sc.zero_grad) *reinterpret_cast<f32x4*>(grad + o[it]) = f32x4{0.f, 0.f, 0.f, 0.f};
      typename Vec4<T>::type q;
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(p[it][k]);
      *reinterpret_cast<typename Vec4<T>::type*>(&tile[r][c4]) = q;
      *reinterpret_cast<typename Vec4<T>::type*>(dst + (o[it] - (long long)j.off)) = q;
    }
  }
  __syncthreads();
  // transposed copy: 256 rows (source columns) x 64 elements = 128 B each; thread: 4 source rows x one column -> 8 bytes
  T* dstT = (T*)j.dstT;
  const int r4 = (t & 15) * 4;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int c = it * 16 + (t >> 4);
    typename Vec4<T>::type q;
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = tile[r4 + k][c];
    *reinterpret_cast<typename Vec4<T>::type*>(dstT + (long long)(bx + c) * j.ldT + by + r4) = q;
  }
}

// every other parameter: the segment table of mra_qformer_load_flat (dst in its stored dtype)
__global__ void __launch_bounds__(256) adam_flat_kernel(float* master, float* grad, float* mom, float* var, const FlatSeg* segs, AdamScalars sc) {
  const FlatSeg sg = segs[blockIdx.x];
  for (int i0 = threadIdx.x * 4; i0 < sg.n; i0 += 4096) {   // four float4 per array in flight per thread
    f32x4 p[4], g[4], m[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * 1024, sg.n - 4);           // clamped: a tail piece is loaded twice, stored once
      const long long o = (long long)sg.src_off + i;
      p[u] = *reinterpret_cast<const f32x4*>(master + o); g[u] = *reinterpret_cast<const f32x4*>(grad + o);
      m[u] = *reinterpret_cast<const f32x4*>(mom + o); v[u] = *reinterpret_cast<const f32x4*>(var + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024;
      if (i >= sg.n) break;
      const long long o = (long long)sg.src_off + i;
      adam_update(p[u], g[u], m[u], v[u], sc);
      *reinterpret_cast<f32x4*>(master + o) = p[u];
      *reinterpret_cast<f32x4*>(mom + o) = m[u];
      *reinterpret_cast<f32x4*>(var + o) = v[u];
      if (sc.zero_grad) *reinterpret_cast<f32x4*>(grad + o) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sg.dtype == 0) {
        *reinterpret_cast<f32x4*>((float*)sg.dst + i) = p[u];
      } else if (sg.dtype == 1) {
        typename Vec4<f16>::type q;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (f16)p[u][k];
        *reinterpret_cast<typename Vec4<f16>::type*>((f16*)sg.dst + i) = q;
      } else {
        typename Vec4<bf16>::type q;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (bf16)p[u][k];
        *reinterpret_cast<typename Vec4<bf16>::type*>((bf16*)sg.dst + i) = q;
      }
    }
  }
}

__global__ void __launch_bounds__(256) axpy_kernel(const float* x, float* y, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 4), b = *reinterpret_cast<f32x4*>(y + i * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) = a + b;
  }
}

}  // namespace

int launch_ln_bwd2(const LnBwdArgs& a, const LnBwdArgs* b, int H, int op_dtype, hipStream_t stream) {
  const int rows_b = b ? b->rows : 0;
  if (a.rows <= 0 && rows_b <= 0) return 0;
  if (a.rows <= 0) return launch_ln_bwd2(*b, nullptr, H, op_dtype, stream);
  if (H % 256 || H > 1024 || H <= 0) return -1;
  if (b && (b->dgamma != nullptr) != (a.dgamma != nullptr)) return -1;   // the kernel's tail branches on the first job's pointer being set per job: keep them alike
  const int wg_a = (a.rows + 15) / 16, wg_b = rows_b > 0 ? (rows_b + 15) / 16 : 0;
  const dim3 grid(wg_a + wg_b), block(256);
  const LnBwdArgs& bb = rows_b > 0 ? *b : a;
#define MRA_LNB(T, NV) hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), grid, block, 0, stream, a, bb, wg_a)
  const int nv = H / 256;
  if (op_dtype == OP_F16) {
    if (nv == 1) MRA_LNB(f16, 1); else if (nv == 2) MRA_LNB(f16, 2); else if (nv == 3) MRA_LNB(f16, 3); else MRA_LNB(f16, 4);
  } else {
    if (nv == 1) MRA_LNB(bf16, 1); else if (nv == 2) MRA_LNB(bf16, 2); else if (nv == 3) MRA_LNB(bf16, 3); else MRA_LNB(bf16, 4);
  }
#undef MRA_LNB
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_ln_bwd(const LnBwdArgs& a, int H, int op_dtype, hipStream_t stream) { return launch_ln_bwd2(a, nullptr, H, op_dtype, stream); }

int launch_gelu(const void* u, const void* df, void* out, long long n, int backward, int op_dtype, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n % 8) return -1;
  const long long n8 = n / 8;
  const unsigned blocks = (unsigned)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  if (op_dtype == OP_F16) {
    if (backward) hipLaunchKernelGGL((gelu_kernel<f16, true>), dim3(blocks), dim3(256), 0, stream, (const f16*)u, (const f16*)df, (f16*)out, n8);
    else hipLaunchKernelGGL((gelu_kernel<f16, false>), dim3(blocks), dim3(256), 0, stream, (const f16*)u, (const f16*)df, (f16*)out, n8);
  } else {
    if (backward) hipLaunchKernelGGL((gelu_kernel<bf16, true>), dim3(blocks), dim3(256), 0, stream, (const bf16*)u, (const bf16*)df, (bf16*)out, n8);
    else hipLaunchKernelGGL((gelu_kernel<bf16, false>), dim3(blocks), dim3(256), 0, stream, (const bf16*)u, (const bf16*)df, (bf16*)out, n8);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

size_t attn_bwd_lds_bytes(int q_rows) {   // the fp32 VALU kernel (kept for A/B: attn_bwd_force_valu)
  const int nqb = (q_rows + QB - 1) / QB;
  return sizeof(float) * ((size_t)nqb * QB * PADW + 2 * QB * PADW + 4 * KB * PADW + 2 * QB * (KB + 1) + 2 * nqb * QB);
}

#ifdef MRA_GEMM_EXPERIMENTS
static int g_attn_bwd_valu = 0;
void attn_bwd_force_valu(int on) { g_attn_bwd_valu = on; }
#else
constexpr int g_attn_bwd_valu = 0;   // the shipped library: the MFMA kernel, no switch
#endif

int launch_attn_bwd(const AttnBwdArgs& a, int op_dtype, hipStream_t stream) {
  if (a.items <= 0 || a.heads <= 0 || a.q_rows <= 0 || a.kv_len <= 0 || !a.lse) return -1;
  if (g_attn_bwd_valu) {
    const size_t lds = attn_bwd_lds_bytes(a.q_rows);
    if (lds > 160 * 1024) return -1;
    void (*kfn)(const AttnBwdArgs) = op_dtype == OP_F16 ? attn_bwd_kernel<f16> : attn_bwd_kernel<bf16>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
    hipLaunchKernelGGL(kfn, dim3(a.items * a.heads), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -4;
  }
  // 16-byte operand rows: every leading dimension / stride a multiple of 8 elements
  if ((a.q_ld | a.o_ld | a.dq_ld | a.k_ld | a.v_ld) & 7) return -1;
  if ((a.q_item_stride | a.o_item_stride | a.dq_item_stride | a.k_item_stride | a.k_head_stride | a.v_item_stride | a.v_head_stride) & 7) return -1;
  const int nqb = (a.q_rows + 31) / 32;
  const size_t lds = 10 * (size_t)bwd::TILE + sizeof(float) * ((size_t)nqb * 32 * 64 + 2 * nqb * 32 + 4 * 32);
  if (lds > 160 * 1024) return -1;
  void (*kfn)(const AttnBwdArgs) = op_dtype == OP_F16 ? attn_bwd_mfma_kernel<f16> : attn_bwd_mfma_kernel<bf16>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
  hipLaunchKernelGGL(kfn, dim3(a.items * a.heads), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_embed_bwd(const float* demb, const long long* ids, int items, int L, int Q, int H, int vocab, float* dquery, float* dpos,
                     float* dword, hipStream_t stream) {
  if (items <= 0) return 0;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(Q + L), dim3(256), 0, stream, demb, ids, items, L, Q, H, vocab, dquery, dpos, dword);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_transpose16_batch(const TrJob* jobs_dev, int njobs, int total_tiles, int op_dtype, hipStream_t stream) {
  if (njobs <= 0 || total_tiles <= 0) return 0;
  if (op_dtype == OP_F16) hipLaunchKernelGGL(transpose16_batch_kernel<f16>, dim3(total_tiles), dim3(256), 0, stream, jobs_dev, njobs);
  else hipLaunchKernelGGL(transpose16_batch_kernel<bf16>, dim3(total_tiles), dim3(256), 0, stream, jobs_dev, njobs);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_transpose16(const void* src, void* dst, int R, int C, int op_dtype, hipStream_t stream) {
  if (R <= 0 || C <= 0) return -1;
  const dim3 grid((C + 31) / 32, (R + 31) / 32), block(256);
  if (op_dtype == OP_F16) hipLaunchKernelGGL(transpose16_kernel<f16>, grid, block, 0, stream, (const f16*)src, (f16*)dst, R, C);
  else hipLaunchKernelGGL(transpose16_kernel<bf16>, grid, block, 0, stream, (const bf16*)src, (bf16*)dst, R, C);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_adam_mats(float* master, float* grad, float* m, float* v, const AdamMatJob* jobs_dev, int njobs, int total_tiles, AdamScalars sc, int op_dtype,
                     hipStream_t stream) {
  if (njobs <= 0 || total_tiles <= 0) return 0;
  if (op_dtype == OP_F16) hipLaunchKernelGGL(adam_mat_kernel<f16>, dim3(total_tiles), dim3(256), 0, stream, master, grad, m, v, jobs_dev, njobs, sc);
  else hipLaunchKernelGGL(adam_mat_kernel<bf16>, dim3(total_tiles), dim3(256), 0, stream, master, grad, m, v, jobs_dev, njobs, sc);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_adam_flat(float* master, float* grad, float* m, float* v, const FlatSeg* segs, int nseg, AdamScalars sc, hipStream_t stream) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(nseg), dim3(256), 0, stream, master, grad, m, v, segs, sc);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_add_f32(const float* x, float* y, long long n, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n % 4) return -1;
  const long long n4 = n / 4;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, stream, x, y, n4);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace mra
