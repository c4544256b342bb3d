// HBM-bound row kernels: LayerNorms, embeddings, dtype conversion.  One wave per row, 16-byte
// accesses, two-pass (mean, then centred variance) statistics in fp32 held in registers.
//
//   ln_rows_kernel       BERT LayerNorm (eps 1e-12) after every residual add of the Q-Former
//                        (HF modeling_instructblip.py:519-530,573-587); writes the fp32 residual
//                        stream and the op-dtype copy the next GEMM reads.
//   modality_ln_kernel   reference models/xinstructblip.py:822-828 (fp32 LayerNorm, eps 1e-5) fused
//                        with the sample-major gather of models/xinstructblip.py:281-285.
//   embed_ln_kernel      Q-Former embeddings (HF modeling_instructblip.py:728-757): queries as is,
//                        text = word + absolute position, then LayerNorm.
#include <algorithm>

#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

__device__ __forceinline__ long long vrow(const RowView& v, int m) {
  const int item = m / v.rpi;
  return (long long)item * v.item_stride + (long long)(m - item * v.rpi) * v.ld;
}

template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v) {
  typename Vec4<T>::type o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
  *reinterpret_cast<typename Vec4<T>::type*>(p) = o;
}

// NV = float4 per lane; H = 256 * NV
template <typename T, int NV>
__global__ void __launch_bounds__(256) ln_rows_kernel(const float* x, RowView xv, int rows, const float* gain,
                                                      const float* bias, float eps, float* y32, RowView y32v, T* y16,
                                                      RowView y16v, const float* gain2, const float* bias2, int period,
                                                      int split, int lane_rows = 0x7fffffff, const float* gain3 = nullptr,
                                                      const float* bias3 = nullptr, const float* gain4 = nullptr,
                                                      const float* bias4 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  // two parameter sets in one launch: rows [split, period) of every item (the text rows) take the second one; rows from lane_rows on
  // (the second Q-Former of a pair forward) take sets 3 / 4 the same way
  if (row >= lane_rows) { gain = gain3; bias = bias3; gain2 = gain4; bias2 = bias4; }
  if (gain2 && row % period >= split) { gain = gain2; bias = bias2; }
  constexpr int H = NV * 256;
  const float* xr = x + vrow(xv, row);
  f32x4 v[NV], g[NV], b[NV];
  // gain and bias ride the same round trip as the row (read next to their use, behind the o32 / o16 branches, each of the NV
  // pieces cost its own)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
    g[i] = *reinterpret_cast<const f32x4*>(gain + (i * 64 + lane) * 4);
    b[i] = *reinterpret_cast<const f32x4*>(bias + (i * 64 + lane) * 4);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[i][e] -= mean; q += v[i][e] * v[i][e]; }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + eps);
  float* o32 = y32 ? y32 + vrow(y32v, row) : nullptr;
  T* o16 = y16 ? y16 + vrow(y16v, row) : nullptr;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = v[i][e] * rstd * g[i][e] + b[i][e];
    if (o32) *reinterpret_cast<f32x4*>(o32 + c) = y;
    if (o16) store4<T>(o16 + c, y);
  }
}

template <typename TI>
__device__ __forceinline__ f32x4 load4(const TI* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 load4<f16>(const f16* p) {
  const Vec4<f16>::type h = *reinterpret_cast<const Vec4<f16>::type*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <>
__device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  const Vec4<bf16>::type h = *reinterpret_cast<const Vec4<bf16>::type*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// 8 consecutive elements of a row as f32 (two f32x4)
struct F8 { f32x4 lo, hi; };
template <typename TI>
__device__ __forceinline__ F8 load8(const TI* p);
template <>
__device__ __forceinline__ F8 load8<float>(const float* p) {
  return F8{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
}
template <>
__device__ __forceinline__ F8 load8<f16>(const f16* p) {
  const Vec8<f16>::type h = *reinterpret_cast<const Vec8<f16>::type*>(p);
  return F8{f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]}};
}
template <>
__device__ __forceinline__ F8 load8<bf16>(const bf16* p) {
  const Vec8<bf16>::type h = *reinterpret_cast<const Vec8<bf16>::type*>(p);
  return F8{f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]}};
}

// rows of E (multiple of 8, <= 512 * MAXC) elements; out item i <- in item index[i].
// One wave per row, 8 elements (16 B of f16) per lane and step.  Every lane loads on every step (a
// lane past the row end re-reads the last chunk and zeroes it): a load under a lane-dependent
// branch would make hipcc wait vmcnt(0) per load and serialise the row (1.7 ms -> 0.3 ms here).
template <typename TI, typename TO, int MAXC>
__global__ void __launch_bounds__(256) modality_ln_kernel(const TI* x, const long long* index, int items, int tokens,
                                                          int E, const float* gain, const float* bias, float eps,
                                                          TO* out) {
  const int lane = threadIdx.x & 63;
  const unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (row >= (unsigned)items * (unsigned)tokens) return;
  const unsigned item = row / (unsigned)tokens;
  const unsigned tok = row - item * (unsigned)tokens;
  const long long src_item = index ? index[item] : (long long)item;
  const TI* xr = x + (src_item * tokens + tok) * E;
  const int nc = E >> 3;
  F8 v[MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    v[i] = load8<TI>(xr + (c < nc ? c : nc - 1) * 8);
    if (c >= nc) { v[i].lo = f32x4{0.f, 0.f, 0.f, 0.f}; v[i].hi = v[i].lo; }
    s += ((v[i].lo[0] + v[i].lo[1]) + (v[i].lo[2] + v[i].lo[3])) + ((v[i].hi[0] + v[i].hi[1]) + (v[i].hi[2] + v[i].hi[3]));
  }
  const float mean = wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const bool live = i * 64 + lane < nc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i].lo[e] = live ? v[i].lo[e] - mean : 0.f;
      v[i].hi[e] = live ? v[i].hi[e] - mean : 0.f;
      q += v[i].lo[e] * v[i].lo[e] + v[i].hi[e] * v[i].hi[e];
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
  TO* orow = out + (long long)row * E;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    const int cc = c < nc ? c : nc - 1;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gain + cc * 8), g1 = *reinterpret_cast<const f32x4*>(gain + cc * 8 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + cc * 8), b1 = *reinterpret_cast<const f32x4*>(bias + cc * 8 + 4);
    typename Vec8<TO>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = from_f32<TO>(v[i].lo[e] * rstd * g0[e] + b0[e]);
      o[4 + e] = from_f32<TO>(v[i].hi[e] * rstd * g1[e] + b1[e]);
    }
    if (c < nc) *reinterpret_cast<typename Vec8<TO>::type*>(orow + c * 8) = o;
  }
}

template <typename T, int NV>
__global__ void __launch_bounds__(256) embed_ln_kernel(const long long* ids, int items, int L, int Q, int vocab,
                                                       const float* query, long long qstride, const float* word, const float* pos,
                                                       const float* gain, const float* bias, float eps, float* h32,
                                                       T* h16, float* pre32) {
  constexpr int H = NV * 256;
  const int lane = threadIdx.x & 63;
  const int S = Q + L;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)items * S) return;
  const int item = (int)(row / S);
  const int s = (int)(row - (long long)item * S);
  f32x4 v[NV];
  float sum = 0.f;
  if (s < Q) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(query + item * qstride + (long long)s * H + (i * 64 + lane) * 4);
  } else {
    long long id = ids[(long long)item * L + (s - Q)];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never fault on a bad token id
    const float* wr = word + id * H;
    const float* pr = pos + (long long)(s - Q) * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      v[i] = *reinterpret_cast<const f32x4*>(wr + c) + *reinterpret_cast<const f32x4*>(pr + c);
    }
  }
  if (pre32) {  // training: the pre-LayerNorm embedding row is needed by the backward
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(pre32 + row * H + (i * 64 + lane) * 4) = v[i];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(sum) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[i][e] -= mean; q += v[i][e] * v[i][e]; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = v[i][e] * rstd * g[e] + b[e];
    *reinterpret_cast<f32x4*>(h32 + row * H + c) = y;
    store4<T>(h16 + row * H + c, y);
  }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) convert_kernel(const TI* src, TO* dst, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = load4<TI>(src + i * 4);
    if constexpr (sizeof(TO) == 4) {
      *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    } else {
      store4<TO>(dst + i * 4, v);
    }
  }
}

__global__ void __launch_bounds__(256) convert_flat_kernel(const float* master, const FlatSeg* segs) {
  const FlatSeg sg = segs[blockIdx.x];
  const float* src = master + sg.src_off;
  for (int i = threadIdx.x * 4; i < sg.n; i += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
    if (sg.dtype == 0) *reinterpret_cast<f32x4*>((float*)sg.dst + i) = v;
    else if (sg.dtype == 1) store4<f16>((f16*)sg.dst + i, v);
    else store4<bf16>((bf16*)sg.dst + i, v);
  }
}

__global__ void __launch_bounds__(256) copy_rows_kernel(const float* src, RowView sv, float* dst, RowView dv, int rows,
                                                        int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = src + vrow(sv, row);
  float* d = dst + vrow(dv, row);
  for (int c = lane * 4; c < H; c += 256) *reinterpret_cast<f32x4*>(d + c) = *reinterpret_cast<const f32x4*>(s + c);
}

// Split-precision cross-attention (mra_qformer_set_cross_precision): an fp32 value x leaves as the operand-dtype pair hi = op(x),
// lo = op(x - hi) (~22 significant bits with f16).  Products of two such pairs run on the ordinary GEMM kernels by concatenation along
// K:  x . w ~ xh wh + xl wh + xh wl = [xh | xl | xh] . [wh | wh | wl]^T.
// Activations: fp32 rows (row view) of C columns, in chunks of `chunk` columns -> dst[row][c / chunk][part][c % chunk], parts
// (hi, lo, hi) for PARTS = 3 or (hi, lo) for PARTS = 2; dst rows are dense (C / chunk * PARTS * chunk elements).
template <typename T, int PARTS>
__global__ void __launch_bounds__(256) split_rows_kernel(const float* src, RowView sv, int rows, int C, int chunk, T* dst) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = src + vrow(sv, row);
  T* d = dst + (long long)row * C * PARTS;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(s + c);
    f32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (float)from_f32<T>(x[e]);
      lo[e] = x[e] - hi[e];
    }
    T* o = d + (long long)(c / chunk) * chunk * PARTS + (c % chunk);   // chunk % 4 == 0: the four values share a chunk
    store4<T>(o, hi);
    store4<T>(o + chunk, lo);
    if constexpr (PARTS == 3) store4<T>(o + 2 * chunk, hi);
  }
}

// Weights (run once per weight change): fp32 W [rows][C] -> [rows][3 C] = (hi | hi | lo) per row
template <typename T>
__global__ void __launch_bounds__(256) split_weight_kernel(const float* W, long long n, int C, T* dst) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const float x = W[i];
    const T hi = from_f32<T>(x);
    const T lo = from_f32<T>(x - (float)hi);
    T* o = dst + r * 3 * C + c;
    o[0] = hi; o[C] = hi; o[2 * C] = lo;
  }
}

// Key weights for the folded form: fp32 W_k [heads * 64][E] -> [heads][E][192] = per (head, e): (hi | hi | lo) over the 64 head dims
template <typename T>
__global__ void __launch_bounds__(256) split_key_weight_kernel(const float* W, int heads, int E, T* dst) {
  const long long n = (long long)heads * 64 * E;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % E);
    const long long hd = i / E;            // head * 64 + d
    const int d = (int)(hd & 63);
    const long long head = hd >> 6;
    const float x = W[i];
    const T hi = from_f32<T>(x);
    const T lo = from_f32<T>(x - (float)hi);
    T* o = dst + (head * E + e) * 192 + d;
    o[0] = hi; o[64] = hi; o[128] = lo;
  }
}

template <typename TI, typename TO>
int modality_ln_t(const void* x, const long long* index, int items, int tokens, int E, const float* gain,
                  const float* bias, float eps, void* out, hipStream_t stream) {
  const long long rows = (long long)items * tokens;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (E <= 1024) {
    hipLaunchKernelGGL((modality_ln_kernel<TI, TO, 2>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, index, items,
                       tokens, E, gain, bias, eps, (TO*)out);
  } else if (E <= 1536) {
    hipLaunchKernelGGL((modality_ln_kernel<TI, TO, 3>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, index, items,
                       tokens, E, gain, bias, eps, (TO*)out);
  } else if (E <= 4096) {
    hipLaunchKernelGGL((modality_ln_kernel<TI, TO, 8>), dim3(blocks), dim3(256), 0, stream, (const TI*)x, index, items,
                       tokens, E, gain, bias, eps, (TO*)out);
  } else {
    return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename TO>
int modality_ln_o(const void* x, int x_dtype, const long long* index, int items, int tokens, int E, const float* gain,
                  const float* bias, float eps, void* out, hipStream_t stream) {
  switch (x_dtype) {
    case 0: return modality_ln_t<float, TO>(x, index, items, tokens, E, gain, bias, eps, out, stream);
    case 1: return modality_ln_t<f16, TO>(x, index, items, tokens, E, gain, bias, eps, out, stream);
    case 2: return modality_ln_t<bf16, TO>(x, index, items, tokens, E, gain, bias, eps, out, stream);
  }
  return -2;
}

template <typename TI>
int convert_i(const void* src, void* dst, int dst_dtype, long long n, hipStream_t stream) {
  const long long n4 = n / 4;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  if (blocks == 0) return 0;
  switch (dst_dtype) {
    case 0: hipLaunchKernelGGL((convert_kernel<TI, float>), dim3(blocks), dim3(256), 0, stream, (const TI*)src, (float*)dst, n4); break;
    case 1: hipLaunchKernelGGL((convert_kernel<TI, f16>), dim3(blocks), dim3(256), 0, stream, (const TI*)src, (f16*)dst, n4); break;
    case 2: hipLaunchKernelGGL((convert_kernel<TI, bf16>), dim3(blocks), dim3(256), 0, stream, (const TI*)src, (bf16*)dst, n4); break;
    default: return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace

int launch_ln_rows(const float* x, RowView xv, int rows, int H, const float* gain, const float* bias, float eps,
                   float* y32, RowView y32v, void* y16, RowView y16v, int op_dtype, hipStream_t stream) {
  return launch_ln_rows2(x, xv, rows, H, gain, bias, nullptr, nullptr, 1, 1, eps, y32, y32v, y16, y16v, op_dtype, stream);
}

int launch_ln_rows2(const float* x, RowView xv, int rows, int H, const float* gain, const float* bias, const float* gain2,
                    const float* bias2, int period, int split, float eps, float* y32, RowView y32v, void* y16, RowView y16v,
                    int op_dtype, hipStream_t stream) {
  return launch_ln_rows4(x, xv, rows, H, gain, bias, gain2, bias2, 0x7fffffff, nullptr, nullptr, nullptr, nullptr, period, split, eps, y32, y32v,
                         y16, y16v, op_dtype, stream);
}

int launch_ln_rows4(const float* x, RowView xv, int rows, int H, const float* gain, const float* bias, const float* gain2, const float* bias2,
                    int lane_rows, const float* gain3, const float* bias3, const float* gain4, const float* bias4, int period, int split,
                    float eps, float* y32, RowView y32v, void* y16, RowView y16v, int op_dtype, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (H % 256 || H > 1024 || H <= 0 || period <= 0) return -1;
  if (lane_rows < rows && (!gain3 || !bias3)) return -1;
  const dim3 grid((rows + 3) / 4), block(256);
#define MRA_LN_CASE(T, NV)                                                                                         \
  hipLaunchKernelGGL((ln_rows_kernel<T, NV>), grid, block, 0, stream, x, xv, rows, gain, bias, eps, y32, y32v, \
                     (T*)y16, y16v, gain2, bias2, period, split, lane_rows, gain3, bias3, gain4, bias4)
  const int nv = H / 256;
  if (op_dtype == OP_F16) {
    if (nv == 1) MRA_LN_CASE(f16, 1); else if (nv == 2) MRA_LN_CASE(f16, 2); else if (nv == 3) MRA_LN_CASE(f16, 3); else MRA_LN_CASE(f16, 4);
  } else {
    if (nv == 1) MRA_LN_CASE(bf16, 1); else if (nv == 2) MRA_LN_CASE(bf16, 2); else if (nv == 3) MRA_LN_CASE(bf16, 3); else MRA_LN_CASE(bf16, 4);
  }
#undef MRA_LN_CASE
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_modality_ln(const void* x, int x_dtype, const long long* item_index, int items, int tokens, int E,
                       const float* gain, const float* bias, float eps, void* out, int op_dtype, hipStream_t stream) {
  if (items <= 0 || tokens <= 0) return 0;
  if (E % 8 || E <= 0) return -1;
  if ((long long)items * tokens > 0x7fffffffLL) return -1;
  return op_dtype == OP_F16
             ? modality_ln_o<f16>(x, x_dtype, item_index, items, tokens, E, gain, bias, eps, out, stream)
             : modality_ln_o<bf16>(x, x_dtype, item_index, items, tokens, E, gain, bias, eps, out, stream);
}

int launch_embed_ln(const long long* ids, int items, int L, int Q, int H, int vocab, const float* query,
                    long long query_item_stride, const float* word, const float* pos, const float* gain, const float* bias, float eps, float* h32,
                    void* h16, float* pre32, int op_dtype, hipStream_t stream) {
  if (items <= 0) return 0;
  if (H % 256 || H > 1024 || H <= 0 || L < 0 || Q < 0) return -1;
  const long long rows = (long long)items * (Q + L);
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define MRA_EM_CASE(T, NV)                                                                                        \
  hipLaunchKernelGGL((embed_ln_kernel<T, NV>), grid, block, 0, stream, ids, items, L, Q, vocab, query, query_item_stride, word, pos, \
                     gain, bias, eps, h32, (T*)h16, pre32)
  const int nv = H / 256;
  if (op_dtype == OP_F16) {
    if (nv == 1) MRA_EM_CASE(f16, 1); else if (nv == 2) MRA_EM_CASE(f16, 2); else if (nv == 3) MRA_EM_CASE(f16, 3); else MRA_EM_CASE(f16, 4);
  } else {
    if (nv == 1) MRA_EM_CASE(bf16, 1); else if (nv == 2) MRA_EM_CASE(bf16, 2); else if (nv == 3) MRA_EM_CASE(bf16, 3); else MRA_EM_CASE(bf16, 4);
  }
#undef MRA_EM_CASE
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_convert(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, hipStream_t stream) {
  if (n % 4) return -1;
  switch (src_dtype) {
    case 0: return convert_i<float>(src, dst, dst_dtype, n, stream);
    case 1: return convert_i<f16>(src, dst, dst_dtype, n, stream);
    case 2: return convert_i<bf16>(src, dst, dst_dtype, n, stream);
  }
  return -2;
}

int launch_convert_flat(const float* master, const FlatSeg* segs, int nseg, hipStream_t stream) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(convert_flat_kernel, dim3(nseg), dim3(256), 0, stream, master, segs);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_split_rows(const float* src, RowView sv, int rows, int C, int chunk, int parts, void* dst, int op_dtype, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (C <= 0 || chunk <= 0 || C % chunk || chunk % 4 || (parts != 2 && parts != 3) || sv.rpi <= 0) return -1;
  const dim3 grid((rows + 3) / 4), block(256);
  if (op_dtype == OP_F16) {
    if (parts == 3) hipLaunchKernelGGL((split_rows_kernel<f16, 3>), grid, block, 0, stream, src, sv, rows, C, chunk, (f16*)dst);
    else hipLaunchKernelGGL((split_rows_kernel<f16, 2>), grid, block, 0, stream, src, sv, rows, C, chunk, (f16*)dst);
  } else {
    if (parts == 3) hipLaunchKernelGGL((split_rows_kernel<bf16, 3>), grid, block, 0, stream, src, sv, rows, C, chunk, (bf16*)dst);
    else hipLaunchKernelGGL((split_rows_kernel<bf16, 2>), grid, block, 0, stream, src, sv, rows, C, chunk, (bf16*)dst);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_split_weight(const float* W, int rows, int C, void* dst, int op_dtype, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return -1;
  const long long n = (long long)rows * C;
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 4096);
  if (op_dtype == OP_F16) hipLaunchKernelGGL(split_weight_kernel<f16>, dim3(blocks), dim3(256), 0, stream, W, n, C, (f16*)dst);
  else hipLaunchKernelGGL(split_weight_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, W, n, C, (bf16*)dst);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_split_key_weight(const float* W, int heads, int E, void* dst, int op_dtype, hipStream_t stream) {
  if (heads <= 0 || E <= 0) return -1;
  const long long n = (long long)heads * 64 * E;
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 4096);
  if (op_dtype == OP_F16) hipLaunchKernelGGL(split_key_weight_kernel<f16>, dim3(blocks), dim3(256), 0, stream, W, heads, E, (f16*)dst);
  else hipLaunchKernelGGL(split_key_weight_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, W, heads, E, (bf16*)dst);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_copy_rows_f32(const float* src, RowView sv, float* dst, RowView dv, int rows, int H, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (H % 4) return -1;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, src, sv, dst, dv, rows, H);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace mra
