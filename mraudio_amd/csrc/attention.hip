// Query-KV attention core for the Q-Former: softmax(Q K^T / 8 + mask) V with 64-wide heads.
//
// Used twice per layer on the path the reference drives at models/xinstructblip.py:286-293:
//   * self-attention over the S = 32 + L tokens of every item (additive text-padding mask,
//     LAVIS form (1 - m) * -10000), units = item x head x 32-row query block;
//   * cross-attention of the 32 learned queries onto the Kv encoder tokens of the item
//     (K/V projected once for all cross layers into a head-major cache, [Kv][64] per head).
//
// One wave owns one 32-query block.  KV tiles of 32 tokens are staged by LDS-DMA into a
// wave-private, double-buffered LDS ring (no workgroup barrier in the loop):
//   S^T[tok][q] = K_tile * Q^T            4 x v_mfma_f32_32x32x16 (K rows from LDS, Q in registers)
//   online softmax per query = per lane column (stats live in the lane, one lane<->lane+32 swap)
//   O^T[d][q] += V^T * P^T                4 x v_mfma_f32_32x32x16; P^T is the S^T accumulator
//                                         itself (rows are the contraction index), V^T comes
//                                         from ds_read_b64_tr_b16 transposed reads of [tok][d].
// LDS rows are 128 B: K chunks are XOR-swizzled with (tok >> 1) & 7 for conflict-free
// ds_read_b128, V chunks with ((tok >> 1) & 1) << 2 for conflict-free transposed reads; the
// swizzle is applied to the DMA source address (LDS-DMA writes linearly).
// Long KV (cross attention): the 4 waves of a workgroup take interleaved tiles of one unit and
// merge through LDS; optionally the KV range is also split over workgroups (partials merged by
// attn_combine_kernel) so that 384 (item, head) units still fill 256 CUs.
#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

constexpr int KVT = 32;                 // tokens per tile
constexpr int TILE_B = KVT * 128;       // bytes per K (or V) tile
constexpr int WAVE_LDS = 4 * TILE_B;    // 2 buffers x (K + V)
constexpr int OPAD = 68;                // padded row of the f32 O scratch
constexpr float LOG2E = 1.4426950408889634f;
constexpr int PART_STRIDE = 32 * 64 + 64;  // floats per partial: O[32][64], m[32], l[32]

template <typename T, bool MASKED, int SPLITW>
__global__ void __launch_bounds__(256) attn_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qblocks = (a.q_rows + 31) >> 5;
  const int nunits = a.items * a.heads * qblocks;
  const int ntiles_all = (a.kv_len + KVT - 1) / KVT;

  // ---- which unit / which tile range ----
  int unit, gs;
  if (SPLITW == 1) {
    unit = blockIdx.x * 4 + wave;
    gs = 0;
  } else {
    unit = blockIdx.x / a.nsplit;
    gs = blockIdx.x - unit * a.nsplit;
  }
  const bool active = unit < nunits;
  if (!active) unit = nunits - 1;
  const int qb = unit % qblocks;
  const int head = (unit / qblocks) % a.heads;
  const int item = unit / (qblocks * a.heads);
  const int q0 = qb * 32;

  const int tps = (ntiles_all + a.nsplit - 1) / a.nsplit;  // tiles per grid split
  const int tb = gs * tps;
  const int te = min(tb + tps, ntiles_all);
  const int tstep = SPLITW;
  const int tfirst = tb + (SPLITW == 1 ? 0 : wave);

  char* wl = smem + wave * WAVE_LDS;  // this wave's ring: [buf][K tile | V tile]
  const int mask_pad = ntiles_all * KVT;
  float* wmask = reinterpret_cast<float*>(smem + 4 * WAVE_LDS) + wave * mask_pad;

  const T* Kb = (const T*)a.K + (long long)item * a.k_item_stride + (long long)head * a.k_head_stride;
  const T* Vb = (const T*)a.V + (long long)item * a.v_item_stride + (long long)head * a.v_head_stride;

  // ---- additive mask in log2 units (self attention only) ----
  if (MASKED) {
    for (int i = lane; i < mask_pad; i += 64) {
      float v = -INFINITY;
      if (i < a.kv_len) {
        const long long mv = a.mask ? a.mask[(long long)item * a.mask_ld + i] : 1;
        v = (1.0f - (float)mv) * (-10000.0f * LOG2E);
      }
      wmask[i] = v;
    }
  }

  // ---- Q fragments: B operand, lane (col q = lane & 31, half h) holds Q[q][16 s + 8 h + j] ----
  typename Vec8<T>::type qf[4];
  {
    const int qr = min(q0 + (lane & 31), a.q_rows - 1);
    const T* qp = (const T*)a.Q + (long long)item * a.q_item_stride + (long long)qr * a.q_ld + head * 64 + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const typename Vec8<T>::type*>(qp + 16 * s);
  }

  // ---- per-lane LDS-DMA sources (row = 8 i + (lane >> 3), physical chunk = lane & 7) ----
  const int srow = lane >> 3, sc = lane & 7;
  auto issue = [&](int buf, int t) {
    char* kb = wl + buf * 2 * TILE_B;
    char* vb = kb + TILE_B;
    const int tok0 = t * KVT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + srow;
      const int tok = min(tok0 + row, a.kv_len - 1);
      const int kc = sc ^ ((row >> 1) & 7);
      const int vc = sc ^ (((row >> 1) & 1) << 2);
      glds16((const char*)(Kb + (long long)tok * a.k_ld) + kc * 16, kb + i * 1024);
      glds16((const char*)(Vb + (long long)tok * a.v_ld) + vc * 16, vb + i * 1024);
    }
  };

  // ---- fragment read addresses ----
  // K (A operand): lane (row tok = lane & 31, half h), k-step s: chunk 2 s + h
  int koff[4];
  {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4);
  }
  // V^T (A operand, rows d): 16-lane group g, lane-in-group i = 4 q4 + p reads 8 bytes of token row
  // T0 + q4 at d = 32 mt + 16 (g & 1) + 4 p; T0 = 16 s + 4 h (+ 8 for elements 4..7).  The V
  // swizzle bit ((row >> 1) & 1) equals (q4 >> 1) & 1 for every (s, e), so each d-half mt needs one
  // per-lane base and the (s, e) step is an immediate: 2048 s + 1024 e.
  unsigned vlane[2];
  {
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, p = i & 3, h = lane >> 5;
    const int row = 4 * h + q4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int c = 4 * mt + 2 * (g & 1) + (p >> 1);
      const int pc = c ^ (((q4 >> 1) & 1) << 2);
      vlane[mt] = row * 128 + pc * 16 + (p & 1) * 8;
    }
  }
  const unsigned wl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wl;

  const float sl2 = a.scale * LOG2E;
  float m_run = -1e30f, l_run = 0.f;
  f32x16 ot[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { ot[0][i] = 0.f; ot[1][i] = 0.f; }

  const int h4 = 4 * (lane >> 5);
  int buf = 0;
  if (tfirst < te) issue(0, tfirst);
  for (int t = tfirst; t < te; t += tstep) {
    const bool more = t + tstep < te;
    if (more) {
      issue(buf ^ 1, t + tstep);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* kb = wl + buf * 2 * TILE_B;
    const char* vb = kb + TILE_B;

    // S^T = K * Q^T
    f32x16 st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) st = mfma32<T>(lds_read8<T>(kb + koff[s]), qf[s], st);

    // scores in log2 units, tail / padding mask
    const int tok0 = t * KVT;
    float mx = -INFINITY;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 madd;
      if (MASKED) {
        madd = *reinterpret_cast<const f32x4*>(wmask + tok0 + 8 * g4 + h4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g4 + e;
        float y = st[i] * sl2;
        if (MASKED) {
          y += madd[e];
        } else {
          if (tok0 + 8 * g4 + h4 + e >= a.kv_len) y = -INFINITY;
        }
        st[i] = y;
        mx = fmaxf(mx, y);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(st[i] - m_new);
      st[i] = p;
      psum += p;
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ot[0][i] *= alpha; ot[1][i] *= alpha; }

    // P^T fragments (B operand): k-step s uses accumulator registers 8 s .. 8 s + 7
    typename Vec8<T>::type pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[s][j] = from_f32<T>(st[8 * s + j]);

    // O^T += V^T * P^T.  The transposed reads go through inline asm: as compiler-visible LDS
    // loads hipcc orders them behind every pending LDS-DMA (s_waitcnt vmcnt(0)), which would
    // drain the prefetch of the next tile.  Loads and their wait sit in one statement.
    {
      const unsigned vaddr0 = wl_lds + buf * 2 * TILE_B + TILE_B + vlane[0];
      const unsigned vaddr1 = wl_lds + buf * 2 * TILE_B + TILE_B + vlane[1];
      i16x4 r00, r01, r02, r03, r10, r11, r12, r13;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\t"
          "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
          "ds_read_b64_tr_b16 %2, %8 offset:2048\n\t"
          "ds_read_b64_tr_b16 %3, %8 offset:3072\n\t"
          "ds_read_b64_tr_b16 %4, %9\n\t"
          "ds_read_b64_tr_b16 %5, %9 offset:1024\n\t"
          "ds_read_b64_tr_b16 %6, %9 offset:2048\n\t"
          "ds_read_b64_tr_b16 %7, %9 offset:3072\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(r00), "=&v"(r01), "=&v"(r02), "=&v"(r03), "=&v"(r10), "=&v"(r11), "=&v"(r12), "=&v"(r13)
          : "v"(vaddr0), "v"(vaddr1)
          : "memory");
      auto cat = [](i16x4 lo, i16x4 hi) {
        i16x8 v8;
        v8[0] = lo[0]; v8[1] = lo[1]; v8[2] = lo[2]; v8[3] = lo[3];
        v8[4] = hi[0]; v8[5] = hi[1]; v8[6] = hi[2]; v8[7] = hi[3];
        return __builtin_bit_cast(typename Vec8<T>::type, v8);
      };
      ot[0] = mfma32<T>(cat(r00, r01), pf[0], ot[0]);
      ot[0] = mfma32<T>(cat(r02, r03), pf[1], ot[0]);
      ot[1] = mfma32<T>(cat(r10, r11), pf[0], ot[1]);
      ot[1] = mfma32<T>(cat(r12, r13), pf[1], ot[1]);
    }
    buf ^= 1;
  }

  // ---- merge ----
  // every wave parks (O unnormalised, m, l) in its own LDS region (the ring is idle now)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  float* ow = reinterpret_cast<float*>(wl);
  float* mw = ow + 32 * OPAD;
  float* lw = mw + 32;
  {
    const int q = lane & 31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v = {ot[mt][4 * g4], ot[mt][4 * g4 + 1], ot[mt][4 * g4 + 2], ot[mt][4 * g4 + 3]};
        *reinterpret_cast<f32x4*>(ow + q * OPAD + 32 * mt + 8 * g4 + h4) = v;
      }
    if (lane < 32) { mw[q] = m_run; lw[q] = l_tot; }
  }
  __syncthreads();

  constexpr int NW = SPLITW;                  // waves merged into one result
  const int tl = SPLITW == 1 ? lane : tid;    // thread index inside the merging group
  constexpr int NTHR = SPLITW == 1 ? 64 : 256;
  const char* gbase = SPLITW == 1 ? wl : smem;
  for (int idx = tl; idx < 256; idx += NTHR) {
    const int q = idx >> 3, dc = idx & 7;
    float M = -1e30f;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, reinterpret_cast<const float*>(gbase + w * WAVE_LDS)[32 * OPAD + q]);
    float L = 0.f;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* rw = reinterpret_cast<const float*>(gbase + w * WAVE_LDS);
      const float wgt = __builtin_amdgcn_exp2f(rw[32 * OPAD + q] - M);
      L += wgt * rw[32 * OPAD + 32 + q];
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(rw + q * OPAD + 8 * dc);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(rw + q * OPAD + 8 * dc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] += wgt * x0[e]; o[4 + e] += wgt * x1[e]; }
    }
    if (!active) continue;
    if (a.lse && a.nsplit == 1 && dc == 0 && q0 + q < a.q_rows)
      a.lse[((long long)item * a.heads + head) * a.q_rows + q0 + q] = M + __builtin_amdgcn_logf(L);  // v_log_f32 = log2
    if (a.nsplit > 1) {
      float* pp = a.part + ((long long)unit * a.nsplit + gs) * PART_STRIDE;
      *reinterpret_cast<f32x4*>(pp + q * 64 + 8 * dc) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(pp + q * 64 + 8 * dc + 4) = f32x4{o[4], o[5], o[6], o[7]};
      if (dc == 0) { pp[32 * 64 + q] = M; pp[32 * 64 + 32 + q] = L; }
    } else if (q0 + q < a.q_rows) {
      const float inv = 1.0f / L;
      typename Vec8<T>::type r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = from_f32<T>(o[e] * inv);
      T* op = (T*)a.O + (long long)item * a.o_item_stride + (long long)(q0 + q) * a.o_ld + head * 64 + 8 * dc;
      *reinterpret_cast<typename Vec8<T>::type*>(op) = r;
    }
  }
}

// merges the grid-split partials of one unit: 256 threads, thread (q = tid >> 3, dc = tid & 7)
template <typename T>
__global__ void __launch_bounds__(256) attn_combine_kernel(const AttnArgs a) {
  const int unit = blockIdx.x;
  const int qblocks = (a.q_rows + 31) >> 5;
  const int qb = unit % qblocks;
  const int head = (unit / qblocks) % a.heads;
  const int item = unit / (qblocks * a.heads);
  const int q = threadIdx.x >> 3, dc = threadIdx.x & 7;
  const float* pp = a.part + (long long)unit * a.nsplit * PART_STRIDE;
  float M = -1e30f;
  for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, pp[s * PART_STRIDE + 32 * 64 + q]);
  float L = 0.f;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.nsplit; ++s) {
    const float* ps = pp + s * PART_STRIDE;
    const float wgt = __builtin_amdgcn_exp2f(ps[32 * 64 + q] - M);
    L += wgt * ps[32 * 64 + 32 + q];
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(ps + q * 64 + 8 * dc);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(ps + q * 64 + 8 * dc + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] += wgt * x0[e]; o[4 + e] += wgt * x1[e]; }
  }
  const int q0 = qb * 32;
  if (q0 + q >= a.q_rows) return;
  if (a.lse && dc == 0) a.lse[((long long)item * a.heads + head) * a.q_rows + q0 + q] = M + __builtin_amdgcn_logf(L);
  const float inv = 1.0f / L;
  typename Vec8<T>::type r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = from_f32<T>(o[e] * inv);
  T* op = (T*)a.O + (long long)item * a.o_item_stride + (long long)(q0 + q) * a.o_ld + head * 64 + 8 * dc;
  *reinterpret_cast<typename Vec8<T>::type*>(op) = r;
}

template <typename T>
int launch_t(const AttnArgs& a, hipStream_t stream) {
  const int qblocks = (a.q_rows + 31) >> 5;
  const int nunits = a.items * a.heads * qblocks;
  const int ntiles = (a.kv_len + KVT - 1) / KVT;
  const bool masked = a.mask != nullptr;
  // short KV: one wave per unit, 4 units per workgroup; long KV: 4 waves share a unit
  const bool shared = ntiles >= 8 && !masked;
  size_t lds = 4 * WAVE_LDS + (masked ? (size_t)4 * ntiles * KVT * sizeof(float) : 0);
  if (lds > 160 * 1024) return -1;
  void (*kfn)(const AttnArgs);
  dim3 grid;
  if (shared) {
    kfn = attn_kernel<T, false, 4>;
    grid = dim3(nunits * a.nsplit);
  } else {
    if (a.nsplit != 1) return -1;
    kfn = masked ? attn_kernel<T, true, 1> : attn_kernel<T, false, 1>;
    grid = dim3((nunits + 3) / 4);
  }
  if (lds > 64 * 1024) {  // (self-attention with very long prompts only; a plain attribute call, not a stream op)
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return -3;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, stream, a);
  if (shared && a.nsplit > 1) {
    hipLaunchKernelGGL(attn_combine_kernel<T>, dim3(nunits), dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace

size_t attn_partial_bytes(int items, int heads, int q_rows, int nsplit) {
  if (nsplit <= 1) return 0;
  const int qblocks = (q_rows + 31) >> 5;
  return (size_t)items * heads * qblocks * nsplit * PART_STRIDE * sizeof(float);
}

int attn_pick_split(int items, int heads, int q_rows, int kv_len) {
  const int qblocks = (q_rows + 31) >> 5;
  const int nunits = items * heads * qblocks;
  const int ntiles = (kv_len + KVT - 1) / KVT;
  if (ntiles < 64) return 1;
  // two 64 KB workgroups fit a CU: aim for a multiple of 512 workgroups with >= 16 tiles each
  int best = 1;
  for (int s = 1; s <= 16; ++s) {
    if (ntiles / s < 16) break;
    best = s;
    if ((long long)nunits * s >= 1536) break;
  }
  return best;
}

int launch_attention(const AttnArgs& a, int op_dtype, hipStream_t stream) {
  if (a.items <= 0 || a.heads <= 0 || a.q_rows <= 0 || a.kv_len <= 0 || a.nsplit < 1) return -1;
  if (a.nsplit > 1 && !a.part) return -1;
  if ((a.q_ld | a.o_ld | a.k_ld | a.v_ld) & 7) return -1;  // 16-byte rows
  return op_dtype == OP_F16 ? launch_t<f16>(a, stream) : launch_t<bf16>(a, stream);
}

}  // namespace mra
