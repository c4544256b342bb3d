// Shared device helpers for the gfx950 kernels (CDNA4, wave64).  HIP only; no host-side types here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mra {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef _Float16 type __attribute__((ext_vector_type(8))); };
template <> struct Vec8<bf16> { typedef __bf16 type __attribute__((ext_vector_type(8))); };
template <typename T> struct Vec4;
template <> struct Vec4<f16> { typedef _Float16 type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<bf16> { typedef __bf16 type __attribute__((ext_vector_type(4))); };

#define MRA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define MRA_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// D(16x16) += A(16x32) * B(32x16); lane l holds A[row l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][col l&15];
// D: col = l&15, row = 4(l>>4)+reg.
template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename Vec8<T>::type a, typename Vec8<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<f16>(Vec8<f16>::type a, Vec8<f16>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<bf16>(Vec8<bf16>::type a, Vec8<bf16>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// D(32x32) += A(32x16) * B(16x32); lane l holds A[row l&31][k=8(l>>5)+j], B[k=8(l>>5)+j][col l&31];
// D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5).
template <typename T>
__device__ __forceinline__ f32x16 mfma32(typename Vec8<T>::type a, typename Vec8<T>::type b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma32<f16>(Vec8<f16>::type a, Vec8<f16>::type b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<bf16>(Vec8<bf16>::type a, Vec8<bf16>::type b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 16-byte async copy global -> LDS.  The LDS destination is wave-uniform base + lane*16
// (the hardware adds the lane part); `lds_wave_base` must be the address lane 0 writes to.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(MRA_GLB_PTR(gsrc), MRA_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// same with the non-temporal cache policy (aux = 2): for bytes ONE workgroup reads once (MI355X_MICROARCH.md, nt-weights)
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(MRA_GLB_PTR(gsrc), MRA_LDS_PTR(lds_wave_base), 16, 0, 2);
}

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type lds_read8(const void* p) {
  return *reinterpret_cast<const typename Vec8<T>::type*>(p);
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements, delivered
// column-major (lane i of the group gets column i, row q in element q).  EXEC must be all ones.
__device__ __forceinline__ i16x4 lds_read_tr4(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p));
}

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ f16 from_f32<f16>(float x) { return (f16)x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 everywhere: far below the f16 / bf16 rounding of every consumer):
// one rcp, one exp2 and five fmas instead of the device library's branchy erff, which cost the fc1 GEMM of the ViT blocks
// a third of its time (1.34 ms against 1.00 ms for the same tile with a plain epilogue, 256 frames)
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float r = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(r, x);
}
// d/du gelu_erf(u) = Phi(u) + u phi(u)
__device__ __forceinline__ float gelu_erf_grad(float u) {
  return 0.5f * (1.0f + erf_fast(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * u * u);
}
// exact (erf) GELU, as BERT's "gelu" (HF ACT2FN["gelu"], reference path uses hidden_act = "gelu")
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

}  // namespace mra
