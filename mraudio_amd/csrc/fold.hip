// Row kernels of the folded cross-attention (mra_abi.hip: cross_fold): with Kv >> hidden it is cheaper to
// move W_k / W_v to the query side than to project every encoder token,
//     scores  S = (Q W_k) enc^T        P = softmax(S / 8)        ctx = (P enc) W_v^T + b_v
// (the key bias adds one constant per query row and cancels in the softmax).  The two big products are
// batched GEMMs on the existing tiles (gemm.hip); this file holds what is left:
//   softmax_rows_kernel   fp32 score rows [rows][ld_s] -> P rows [rows][ld_p] in the operand dtype, padding zeroed
//   transpose_pad_kernel  batched [R][C] -> [C][ld_d] transpose with zeroed padding columns: enc -> enc^T per item
//                         (the K-contiguous B operand of P . enc) and W_k -> per-head [E][64] blocks
// Reference arithmetic: HF modeling_instructblip.py:464-515 (the same softmax(QK^T/8)V, re-associated).
#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

constexpr float LOG2E_F = 1.4426950408889634f;

// One workgroup per row.  NV > 0: the row (kv <= 1024 * NV floats) stays in registers, NV float4 per thread --
// one read, one write; NV == 0: any length, three passes over the row (the later ones hit L2).
template <typename T, int NV>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* S, long long ld_s, T* P, long long ld_p, int kv, int kvp,
                                                           float sl2) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = S + (long long)blockIdx.x * ld_s;
  T* p = P + (long long)blockIdx.x * ld_p;
  constexpr int NR = NV > 0 ? NV : 1;
  f32x4 v[NR];
  float m = -3.0e38f;
  if constexpr (NV > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = 4 * (tid + j * 256);
      if (c + 3 < kv) {
        v[j] = *reinterpret_cast<const f32x4*>(s + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = c + e < kv ? s[c + e] : -3.0e38f;
      }
      m = fmaxf(fmaxf(m, fmaxf(v[j][0], v[j][1])), fmaxf(v[j][2], v[j][3]));
    }
  } else {
    for (int i = tid; i < kv; i += 256) m = fmaxf(m, s[i]);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * sl2;
  float l = 0.f;
  if constexpr (NV > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[j][e] = __builtin_amdgcn_exp2f(v[j][e] * sl2 - m);   // masked tail: exp2(-huge) = 0
        l += v[j][e];
      }
  } else {
    for (int i = tid; i < kv; i += 256) l += __builtin_amdgcn_exp2f(s[i] * sl2 - m);
  }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  // normalised probabilities; columns [kv, kvp) are zeroed (they meet finite enc^T padding in the next GEMM)
  if constexpr (NV > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = 4 * (tid + j * 256);
      if (c < kvp) {
        typename Vec4<T>::type o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[j][e] * inv);
        *reinterpret_cast<typename Vec4<T>::type*>(p + c) = o;
      }
    }
  } else {
    for (int i = tid; i < kvp; i += 256) p[i] = from_f32<T>(i < kv ? __builtin_amdgcn_exp2f(s[i] * sl2 - m) * inv : 0.f);
  }
}

// Second half of the split softmax.  The scores GEMM (EPI_SOFTPART) left P~ = exp2(s - m_tile) per column tile and the
// tile statistics (m_tile, l_tile) of every row; here one workgroup per row finds m_row = max m_tile,
// L = sum exp2(m_tile - m_row) l_tile and rescales the row in place by g_tile = exp2(m_tile - m_row) / L.
// Columns from ntiles * tile_cols to kvp were never written by the GEMM and are zeroed.
template <typename T>
__global__ void __launch_bounds__(256) softmax_rescale_kernel(T* P, long long ld_p, const float* stat_m, const float* stat_l, int ntiles,
                                                              int tile_cols, int kvp) {
  __shared__ float g[128];
  const int tid = threadIdx.x;
  const float* sm = stat_m + (long long)blockIdx.x * ntiles;
  const float* sl = stat_l + (long long)blockIdx.x * ntiles;
  if (tid < 64) {
    float m = -3.0e38f;
    for (int t = tid; t < ntiles; t += 64) m = fmaxf(m, sm[t]);
    m = wave_max(m);
    float l = 0.f;
    for (int t = tid; t < ntiles; t += 64) l += __builtin_amdgcn_exp2f(sm[t] - m) * sl[t];
    l = wave_sum(l);
    const float inv = 1.0f / l;
    for (int t = tid; t < ntiles; t += 64) g[t] = __builtin_amdgcn_exp2f(sm[t] - m) * inv;
  }
  __syncthreads();
  using V8 = typename Vec8<T>::type;
  T* p = P + (long long)blockIdx.x * ld_p;
  const int cpt = tile_cols >> 3;   // 16-byte chunks per tile
  for (int c = tid; c < (kvp >> 3); c += 256) {
    const int t = c / cpt;
    V8 v;
    if (t < ntiles) {
      v = *reinterpret_cast<const V8*>(p + 8 * c);
      const float s = g[t];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = from_f32<T>((float)v[e] * s);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
    }
    *reinterpret_cast<V8*>(p + 8 * c) = v;
  }
}

// The same row statistics as softmax_rescale_kernel, but only the factors leave: one wave per row, factors in the [item][tile][512] layout
// the P . enc GEMM's loader waves copy into LDS one tile slice (2 KB) at a time.
// It also zeroes the row's P~ columns from ntiles * tile_cols to kvp, which the scores GEMM never writes and the P . enc GEMM reads
// (its K runs to kvp): zero times any finite factor is zero, so that GEMM needs no column test of its own.
__global__ void __launch_bounds__(256) fold_rowfactor_kernel(const float* stat_m, const float* stat_l, float* factors, int rows, int R, int ntiles,
                                                             unsigned short* P, long long ld_p, int tile_cols, int kvp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  for (int c = ntiles * tile_cols + lane; c < kvp; c += 64) P[(long long)row * ld_p + c] = 0;
  const float* sm = stat_m + (long long)row * ntiles;
  const float* sl = stat_l + (long long)row * ntiles;
  float m = -3.0e38f;
  for (int t = lane; t < ntiles; t += 64) m = fmaxf(m, sm[t]);
  m = wave_max(m);
  float l = 0.f;
  for (int t = lane; t < ntiles; t += 64) l += __builtin_amdgcn_exp2f(sm[t] - m) * sl[t];
  l = wave_sum(l);
  const float inv = 1.0f / l;
  const int item = row / R, r = row - item * R;
  for (int t = lane; t < ntiles; t += 64) factors[((long long)item * ntiles + t) * 512 + r] = __builtin_amdgcn_exp2f(sm[t] - m) * inv;
}

// dst[b][c][r] = src[b][r][c] for r < R, 0 for R <= r < ld_d; 32 x 32 tiles, grid (ceil(C/32), ceil(ld_d/32), batch)
template <typename T>
__global__ void __launch_bounds__(256) transpose_pad_kernel(const T* src, T* dst, int R, int C, int ld_d, long long src_bs,
                                                            long long dst_bs) {
  __shared__ T tile[32][33];
  src += (long long)blockIdx.z * src_bs;
  dst += (long long)blockIdx.z * dst_bs;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: source column block, by: source row block
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    tile[j][tx] = (by + j < R && bx + tx < C) ? src[(long long)(by + j) * C + bx + tx] : from_f32<T>(0.f);
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < C && by + tx < ld_d) dst[(long long)(bx + j) * ld_d + by + tx] = tile[tx][j];
}

// 64 x 64 tiles with 16-byte global accesses (C % 8 == 0, ld_d % 8 == 0): thread t loads chunk (t & 7) of rows
// (t >> 3) and (t >> 3) + 32, then stores chunk (t & 7) of destination rows (t >> 3), (t >> 3) + 32.
template <typename T>
__global__ void __launch_bounds__(256) transpose_pad64_kernel(const T* src, T* dst, int R, int C, int ld_d, long long src_bs,
                                                              long long dst_bs) {
  __shared__ T tile[64][72];   // 144-byte pitch: 16-byte aligned rows
  src += (long long)blockIdx.z * src_bs;
  dst += (long long)blockIdx.z * dst_bs;
  const int bx = blockIdx.x * 64, by = blockIdx.y * 64;   // bx: source column block, by: source row block
  const int ch = (threadIdx.x & 7) * 8, r0 = threadIdx.x >> 3;
  using V8 = typename Vec8<T>::type;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + 32 * i;
    V8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
    if (by + r < R && bx + ch < C) v = *reinterpret_cast<const V8*>(src + (long long)(by + r) * C + bx + ch);
    *reinterpret_cast<V8*>(&tile[r][ch]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = r0 + 32 * i;       // destination row = source column bx + c
    if (bx + c < C && by + ch < ld_d) {
      V8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[ch + e][c];
      *reinterpret_cast<V8*>(dst + (long long)(bx + c) * ld_d + by + ch) = v;
    }
  }
}

}  // namespace

int launch_softmax_rows(const float* S, long long ld_s, void* P, long long ld_p, int rows, int kv, int kvp, float scale, int op_dtype,
                        hipStream_t stream) {
  if (rows <= 0) return 0;
  if (kv <= 0 || kvp < kv || (kvp & 3) || (ld_s & 3) || (ld_p & 3) || ld_p < kvp || ld_s < kv) return -1;
  const float sl2 = scale * LOG2E_F;
  const int nv = (kvp + 1023) / 1024;   // float4 per thread that cover the padded row
#define MRA_SM(T, NV) hipLaunchKernelGGL((softmax_rows_kernel<T, NV>), dim3(rows), dim3(256), 0, stream, S, ld_s, (T*)P, ld_p, kv, kvp, sl2)
#define MRA_SM_T(T)                                                                                   \
  do {                                                                                                \
    if (nv <= 1) MRA_SM(T, 1); else if (nv <= 2) MRA_SM(T, 2); else if (nv <= 4) MRA_SM(T, 4);          \
    else if (nv <= 9) MRA_SM(T, 9); else if (nv <= 16) MRA_SM(T, 16); else MRA_SM(T, 0);               \
  } while (0)
  if (op_dtype == OP_F16) MRA_SM_T(f16); else MRA_SM_T(bf16);
#undef MRA_SM_T
#undef MRA_SM
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_fold_rowfactor(const float* stat_m, const float* stat_l, float* factors, int rows, int R, int ntiles, void* P, long long ld_p, int tile_cols,
                          int kvp, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (ntiles <= 0 || R <= 0 || R > 512 || rows % R || !P || ld_p < kvp || ntiles * tile_cols > kvp) return -1;
  hipLaunchKernelGGL(fold_rowfactor_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, stat_m, stat_l, factors, rows, R, ntiles, (unsigned short*)P, ld_p,
                     tile_cols, kvp);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_softmax_rescale(void* P, long long ld_p, const float* stat_m, const float* stat_l, int rows, int ntiles, int tile_cols, int kvp,
                           int op_dtype, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (ntiles <= 0 || ntiles > 128 || (tile_cols & 7) || (kvp & 7) || (ld_p & 7) || ld_p < kvp) return -1;
  if (op_dtype == OP_F16) hipLaunchKernelGGL(softmax_rescale_kernel<f16>, dim3(rows), dim3(256), 0, stream, (f16*)P, ld_p, stat_m, stat_l, ntiles, tile_cols, kvp);
  else hipLaunchKernelGGL(softmax_rescale_kernel<bf16>, dim3(rows), dim3(256), 0, stream, (bf16*)P, ld_p, stat_m, stat_l, ntiles, tile_cols, kvp);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_transpose_pad(const void* src, void* dst, int R, int C, int ld_d, long long src_bs, long long dst_bs, int batch, int op_dtype,
                         hipStream_t stream) {
  if (batch <= 0 || R <= 0 || C <= 0) return 0;
  if (ld_d < R || batch > 65535) return -1;
  if (C % 8 == 0 && ld_d % 8 == 0 && C >= 64 && ld_d >= 64) {
    const dim3 grid64((C + 63) / 64, (ld_d + 63) / 64, batch);
    if (op_dtype == OP_F16) hipLaunchKernelGGL(transpose_pad64_kernel<f16>, grid64, dim3(256), 0, stream, (const f16*)src, (f16*)dst, R, C, ld_d, src_bs, dst_bs);
    else hipLaunchKernelGGL(transpose_pad64_kernel<bf16>, grid64, dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, R, C, ld_d, src_bs, dst_bs);
    return hipGetLastError() == hipSuccess ? 0 : -4;
  }
  const dim3 grid((C + 31) / 32, (ld_d + 31) / 32, batch), block(256);
  if (op_dtype == OP_F16) hipLaunchKernelGGL(transpose_pad_kernel<f16>, grid, block, 0, stream, (const f16*)src, (f16*)dst, R, C, ld_d, src_bs, dst_bs);
  else hipLaunchKernelGGL(transpose_pad_kernel<bf16>, grid, block, 0, stream, (const bf16*)src, (bf16*)dst, R, C, ld_d, src_bs, dst_bs);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace mra
