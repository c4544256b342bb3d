// Clip x query cosine-similarity scorer, modality fusion and the integer span pick.
//
// Not in the reference: mrAudio decodes spans by prompting a 7B LLM with the projected query
// embeddings (models/xinstructblip.py:346-397) and parsing "[[s, e]]" text (utils/utils.py:66-132).
// This build's north star replaces that stage with a similarity scorer; its definition is the
// oracle's (oracle/qformer_ref.py cosine_scores / fuse_logits / span_from_logits):
//   sim[n][q] = <z_nq, t_n> / (max(|z_nq|, eps) * max(|t_n|, eps)),  logit[n] = max_q sim[n][q]
// HBM-bound (reads N*Q*H floats once): one wave per (clip, query) row, 16-byte loads,
// xor-shuffle wave reduction; the 32 query rows of a clip sit in one 512-thread workgroup so the
// max over queries is an LDS reduction, no atomics.
#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

// grid = items, block = 512 (8 waves); wave w handles queries w, w + 8, ...
__global__ void __launch_bounds__(512) cosine_kernel(const float* z, const float* t, int t_rows, int Q, int H, float eps,
                                                     float* sim, float* logit) {
  __shared__ float smax[8];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* tr = t + (long long)(t_rows == 1 ? 0 : n) * H;
  float tt = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(tr + c);
    tt += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
  }
  const float tn = fmaxf(sqrtf(wave_sum(tt)), eps);
  float best = -INFINITY;
  for (int q = wave; q < Q; q += 8) {
    const float* zr = z + ((long long)n * Q + q) * H;
    float dot = 0.f, zz = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(zr + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(tr + c);
      dot += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
      zz += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
    }
    dot = wave_sum(dot);
    const float zn = fmaxf(sqrtf(wave_sum(zz)), eps);
    const float s = dot / (zn * tn);
    if (lane == 0 && sim) sim[(long long)n * Q + q] = s;
    best = fmaxf(best, s);
  }
  if (lane == 0) smax[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = smax[0];
    for (int w = 1; w < 8; ++w) m = fmaxf(m, smax[w]);
    logit[n] = m;
  }
}

// out[i] = sum_m logits[m][i] * w[m], accumulated left to right in fp32 (as the oracle does)
struct FuseArgs {
  const float* x[4];
  float w[4];
  int nmod;
};
__global__ void __launch_bounds__(256) fuse_kernel(FuseArgs a, int n, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  {
#pragma clang fp contract(off)  // the oracle rounds the product and the sum separately: no fma here
    for (int m = 0; m < a.nmod; ++m) {
      const float prod = a.x[m][i] * a.w[m];
      acc = acc + prod;
    }
  }
  out[i] = acc;
}

// one 64-lane wave per video; clips <= 4096.  Integer result must equal the oracle's on the same
// fp32 logits: first argmax, thr = lo + alpha * (hi - lo) (two roundings, no fma contraction),
// then grow while the neighbour >= thr.
__global__ void __launch_bounds__(64) span_kernel(const float* logits, int clips, float alpha, int* spans) {
  const int v = blockIdx.x;
  const int lane = threadIdx.x;
  const float* x = logits + (long long)v * clips;
  float hi = -INFINITY, lo = INFINITY;
  int arg = 0x7fffffff;
  for (int i = lane; i < clips; i += 64) {
    const float y = x[i];
    if (y > hi) { hi = y; arg = i; }   // strict > keeps the first index inside a lane's stride
    lo = fminf(lo, y);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ohi = __shfl_xor(hi, o, 64);
    const int oarg = __shfl_xor(arg, o, 64);
    if (ohi > hi || (ohi == hi && oarg < arg)) { hi = ohi; arg = oarg; }
    lo = fminf(lo, __shfl_xor(lo, o, 64));
  }
  if (lane == 0) {
    const float range = hi - lo;
    float thr;
    {
#pragma clang fp contract(off)  // multiply and add rounded separately, as the oracle does
      const float prod = alpha * range;
      thr = lo + prod;
    }
    int s = arg, e = arg;
    while (s - 1 >= 0 && x[s - 1] >= thr) --s;
    while (e + 1 < clips && x[e + 1] >= thr) ++e;
    spans[2 * v] = s;
    spans[2 * v + 1] = e;
  }
}

}  // namespace

int launch_cosine_score(const float* z, const float* t, int t_rows, int items, int Q, int H, float eps, float* sim,
                        float* logit, hipStream_t stream) {
  if (items <= 0) return 0;
  if (H % 4 || Q <= 0 || (t_rows != 1 && t_rows != items)) return -1;
  hipLaunchKernelGGL(cosine_kernel, dim3(items), dim3(512), 0, stream, z, t, t_rows, Q, H, eps, sim, logit);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_fuse_logits(const float* const* logits, const float* weights, int nmod, int n, float* out,
                       hipStream_t stream) {
  if (nmod < 1 || nmod > 4) return -1;
  if (n <= 0) return 0;
  FuseArgs a;
  for (int m = 0; m < 4; ++m) {
    a.x[m] = m < nmod ? logits[m] : nullptr;
    a.w[m] = m < nmod ? (weights ? weights[m] : 1.0f / nmod) : 0.f;
  }
  a.nmod = nmod;
  hipLaunchKernelGGL(fuse_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, n, out);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_span(const float* logits, int videos, int clips, float alpha, int* spans, hipStream_t stream) {
  if (videos <= 0) return 0;
  if (clips <= 0) return -1;
  hipLaunchKernelGGL(span_kernel, dim3(videos), dim3(64), 0, stream, logits, clips, alpha, spans);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace mra
