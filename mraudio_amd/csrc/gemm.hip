// MFMA GEMM for every projection on the Q-Former path:  C[m][n] = sum_k A[m][k] * W[n][k] (+ epilogue).
//
// Replaces the nn.Linear calls inside the LAVIS Q-Former that the reference invokes at
// models/xinstructblip.py:286-293 (self/cross Q,K,V and output projections, both feed-forwards)
// and the llm_proj at models/xinstructblip.py:303.  Both operands are K-contiguous ([rows][K]),
// so the product is computed "swapped": the weight rows ride the MFMA row index and the
// activation rows the MFMA column index (D[n][m]); each lane then owns 4 consecutive n of one
// output row m, i.e. 8 contiguous bytes (f16) / 16 bytes (f32) of C.
//
// Two main loops over the same tiles / epilogues (gfx950, v_mfma_f32_16x16x32_{f16,bf16}, fp32 acc):
//   ring (default)  LDS ring of NS slots, one slot = a 32-deep K slice of both operand tiles
//                   (64-byte rows).  LDS-DMA (global_load_lds, 16 B/lane) keeps NS - SPI slots in
//                   flight at all times behind a COUNTED s_waitcnt vmcnt and one raw s_barrier per
//                   SPI slots, so the per-CU load path (the real bound of a 256x256 tile: 128 flop
//                   per staged byte against ~30 B/clk/CU from L2) never drains.  256x256: 5 slots x
//                   32 KiB = all 160 KiB of LDS, 2 slots (BK 64) consumed per barrier.
//   v1              two 64-deep buffers, vmcnt(0) + __syncthreads per K tile (round-1 first cut,
//                   kept for A/B runs: gemm_force_variant(1)).
// Bank conflicts: rows are 64 B (ring) / 128 B (v1), so the 16-byte chunk index is XOR-swizzled
// ((-(row >> 2)) & 3, resp. (row >> 1) & 7) on the per-lane SOURCE address (LDS-DMA writes linearly)
// and on the ds_read_b128 address: conflict-free for all four 16-lane groups.
// Workgroups walk tiles in 8-row-tile panels after an XCD-contiguous remap so that the 32
// concurrently running workgroups of one XCD share operand panels in that XCD's L2.
#include <mutex>
#include <set>
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
__device__ __forceinline__ const GemmProb& pick_tile(const GemmArgs& args, int& n0, int& m0) {
  int id = blockIdx.x;
  {
    const int nwg = args.total_tiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int g = (args.ngroups > 1 && id >= args.p[1].tile_begin) ? 1 : 0;
  const GemmProb& P = args.p[g];
  const int pid = id - P.tile_begin;
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
template <typename T, int FN, int FM, int EPI>
__device__ __forceinline__ void epilogue(const GemmProb& P, f32x4 (&acc)[FN][FM], int n_base, int m_base, int lane) {
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
      f32x4 v = acc[i][j];
      if (P.bias) v += *reinterpret_cast<const f32x4*>(P.bias + n);
      if constexpr (EPI == EPI_GELU_OP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if constexpr (EPI == EPI_RES_F32) v += *reinterpret_cast<const f32x4*>(P.R + roff + n);
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

// 16-bit outputs (EPI_OP, EPI_GELU_OP, EPI_KV) leave through LDS: a lane's natural store is 8 bytes
// of one output row, 16 rows per wave instruction -- 32-byte fragments of 128-byte lines; measured on
// the K/V projection that direct epilogue cost 23 k of the 105 k cycles of a 256x256 tile (store-issue
// bound, cdna_hip_programming.md T21).  Here the tile is first written to LDS as TN/64 column blocks of
// [TM rows][64 cols] (chunk index XOR (row & 7) against ds_write conflicts), then every thread moves
// 16-byte chunks out in LDS order: one wave instruction = 8 rows x 128 B, whole lines; for the
// head-major K/V cache that is 1 KiB contiguous.  Must be entered after a workgroup barrier (the K
// loop's LDS reads are over); smem needs TN * TM * 2 bytes.
template <typename T, int TN, int TM, int FN, int FM, int NT, int EPI>
__device__ __forceinline__ void epilogue_lds16(const GemmProb& P, f32x4 (&acc)[FN][FM], char* smem, int n0, int m0, int wn0,
                                               int wm0, int tid) {
  const int lane = tid & 63;
  const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int nl = wn0 + i * 16 + ln;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.bias) bv = *reinterpret_cast<const f32x4*>(P.bias + n0 + nl);
    const int hq = nl >> 6, d = nl & 63;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int ml = wm0 + j * 16 + lm;
      f32x4 v = acc[i][j] + bv;
      if constexpr (EPI == EPI_GELU_OP) {
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
    int item = item0, r = r0 + row;
    while (r >= rpi) { r -= rpi; ++item; }
    const int cc = c ^ (row & 7);
    if constexpr (EPI == EPI_KV) rowoff[u] = ((long long)item * P.kv_heads * P.kv_tokens + r) * 64 + cc * 8;
    else rowoff[u] = (long long)item * P.c.item_stride + (long long)r * P.c.ld + cc * 8;
  }
#pragma unroll
  for (int hq = 0; hq < TN / 64; ++hq) {
    const int nb = n0 + hq * 64;
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
      const typename Vec8<T>::type val = *reinterpret_cast<const typename Vec8<T>::type*>(smem + (size_t)q * 16);
      *reinterpret_cast<typename Vec8<T>::type*>((T*)P.C + blk + rowoff[u]) = val;
    }
  }
}

// =================================================================================================
// ring main loop: slots of 32 k (64-byte rows)
// =================================================================================================
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt immediate");
}

template <typename T, int TN, int TM, int WGN, int WGM, int NS, int SPI, int EPI>
__global__ void __launch_bounds__(WGN* WGM * 64) gemm_ring_kernel(const GemmArgs args) {
  constexpr int NT = WGN * WGM * 64;
  constexpr int WTN = TN / WGN, WTM = TM / WGM;
  constexpr int FN = WTN / 16, FM = WTM / 16;
  constexpr int SK = 32, SROW = SK * 2;                  // slot depth, bytes per staged row
  constexpr int IW = TN * 4 / NT, IX = TM * 4 / NT;      // 16-byte chunks per thread per operand per slot
  static_assert(TN * 4 % NT == 0 && TM * 4 % NT == 0, "tile/threads mismatch");
  static_assert(NS >= 2 * SPI, "ring too small");
  constexpr int LPS = IW + IX;                           // LDS-DMA instructions per thread per slot
  constexpr int SLOT = (TN + TM) * SROW;
  constexpr int KEEP = LPS * (NS - 2 * SPI);             // loads that may stay in flight at the wait
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn0 = (wave / WGM) * WTN;
  const int wm0 = (wave % WGM) * WTM;
  int n0, m0;
  const GemmProb& P = pick_tile<TN, TM>(args, n0, m0);
  const int K = P.K, M = P.M;

  // per-lane source pointers; physical chunk c' = q & 3 of row q >> 2 holds source chunk c' ^ ((-(row >> 2)) & 3)
  const char* srcW[IW];
  const char* srcX[IX];
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 2, c = (q & 3) ^ ((-(row >> 2)) & 3);
    srcW[i] = (const char*)P.W + ((long long)(n0 + row) * K + c * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 2, c = (q & 3) ^ ((-(row >> 2)) & 3);
    const int m = min(m0 + row, M - 1);  // rows past M are computed on a clamped row and never stored
    srcX[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
  }
  const int wave_q0 = wave * 64;

  auto issue = [&](int slot, int t) {
    char* base = smem + slot * SLOT;
    const long long koff = (long long)t * SROW;
#pragma unroll
    for (int i = 0; i < IW; ++i) glds16(srcW[i] + koff, base + (wave_q0 + i * NT) * 16);
#pragma unroll
    for (int i = 0; i < IX; ++i) glds16(srcX[i] + koff, base + TN * SROW + (wave_q0 + i * NT) * 16);
  };

  // fragment read offset: row (lane & 15), logical chunk (lane >> 4), swizzled
  const int foff = (lane & 15) * SROW + ((((lane >> 4)) ^ ((-((lane & 15) >> 2)) & 3)) << 4);

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nslots = K / SK;          // host guarantees K % (SK * SPI) == 0
  const int niter = nslots / SPI;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nslots) issue(s, s);

  int cslot = 0;                      // ring position of the first slot of this iteration
  for (int it = 0; it < niter; ++it) {
    // slots it*SPI .. it*SPI+SPI-1 must have landed; younger ones may stay in flight
    if (nslots - (it + 1) * SPI >= NS - 2 * SPI) wait_vmcnt<KEEP>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // the raw barrier is IntrNoMem: keep LDS traffic below it
    // refill the slots iteration it-1 consumed (every wave is past its reads of them now)
    if (it >= 1) {
#pragma unroll
      for (int j = 0; j < SPI; ++j) {
        const int t = (it - 1) * SPI + NS + j;
        int rs = cslot - SPI + j;
        rs = rs < 0 ? rs + NS : rs;
        if (t < nslots) issue(rs, t);
      }
    }
#pragma unroll
    for (int j = 0; j < SPI; ++j) {
      int s = cslot + j;
      s = s >= NS ? s - NS : s;
      const char* wb = smem + s * SLOT + wn0 * SROW + foff;
      const char* xb = smem + s * SLOT + TN * SROW + wm0 * SROW + foff;
      typename Vec8<T>::type a[FN], b[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) a[i] = lds_read8<T>(wb + i * 16 * SROW);
#pragma unroll
      for (int jj = 0; jj < FM; ++jj) b[jj] = lds_read8<T>(xb + jj * 16 * SROW);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int jj = 0; jj < FM; ++jj) acc[i][jj] = mfma16<T>(a[i], b[jj], acc[i][jj]);
      __builtin_amdgcn_s_setprio(0);
    }
    cslot += SPI;
    cslot = cslot >= NS ? cslot - NS : cslot;
  }
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, NT, EPI>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
}

// =================================================================================================
// v1 main loop: two 64-deep buffers (128-byte rows), vmcnt(0) + __syncthreads per K tile
// =================================================================================================
// PF: every wave also touches, two K tiles ahead of the LDS-DMA stage, the 64 cache lines of its share
// of that tile with two 4-byte LDS-DMA loads per lane-line (dummy LDS target), so that the ~22 % of
// stage loads that would miss the XCD's L2 (measured TCC hit rate 78 %) are L2 hits by the time the
// stage is issued: vmcnt retires in order, so one slow miss per stage sets the whole stage's latency.
// SPREAD: the LDS-DMA instructions of the next K tile are not issued in one burst after the barrier
// but one at a time between groups of MFMAs (an LDS-DMA issue costs 60-185 cycles of the wave's issue
// slot; in a burst both waves of a SIMD pay it at the same time and the matrix pipe idles).  The two
// waves that share a SIMD (wave w and w + 4) use opposite phases so their VMEM issues alternate.
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <typename T, int TN, int TM, int WGN, int WGM, int EPI, bool PF, bool SPREAD = false, bool STAMP = false>
__global__ void __launch_bounds__(WGN* WGM * 64) gemm_kernel(const GemmArgs args) {
  constexpr int BK = 64, ROWB = BK * 2;
  constexpr int NT = WGN * WGM * 64;
  constexpr int WTN = TN / WGN, WTM = TM / WGM;
  constexpr int FN = WTN / 16, FM = WTM / 16;
  constexpr int IW = TN * 8 / NT, IX = TM * 8 / NT;
  static_assert(TN * 8 % NT == 0 && TM * 8 % NT == 0, "tile/threads mismatch");
  constexpr int BUF = (TN + TM) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned long long t_entry = STAMP ? stamp() : 0;

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
    const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
    srcW[i] = (const char*)P.W + ((long long)(n0 + row) * K + c * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < IX; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
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
  const int spread_phase = (wave >= (WGN * WGM) / 2) ? 1 : 0;  // waves w and w + half share a SIMD
  // L2 prefetch: line q of a stage = row q of the W tile (q < TN) or row q - TN of the X tile
  const char* pfsrc = nullptr;
  char* pfdst = smem + 2 * BUF + wave * 512;
  if (PF) {
    const int q = tid % (TN + TM);
    if (q < TN) pfsrc = (const char*)P.W + (long long)(n0 + q) * K * 2;
    else pfsrc = (const char*)P.A + view_off(P.a, min(m0 + q - TN, M - 1)) * 2;
  }
  auto prefetch = [&](int kt) {
    __builtin_amdgcn_global_load_lds(MRA_GLB_PTR(pfsrc + (long long)kt * ROWB), MRA_LDS_PTR(pfdst), 4, 0, 0);
    __builtin_amdgcn_global_load_lds(MRA_GLB_PTR(pfsrc + (long long)kt * ROWB), MRA_LDS_PTR(pfdst), 4, 64, 0);
  };
  if (PF) {
    if (1 < nk) prefetch(1);
    if (2 < nk) prefetch(2);
  }
  stage(0, 0);
  unsigned long long tsum[4] = {0, 0, 0, 0}, tq[5];
  const unsigned long long t_loop0 = STAMP ? stamp() : 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (STAMP) {
      tq[0] = stamp();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tq[1] = stamp();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      tq[2] = stamp();
    } else if (PF) {
      // the two youngest operations are the prefetch of tile kt + 2 (if one was issued): leave them in flight
      if (kt >= 1 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const bool more = kt + 1 < nk;
    if (!SPREAD && more) stage((kt + 1) & 1, kt + 1);
    if (STAMP) tq[3] = stamp();
    if (PF && kt + 3 < nk) prefetch(kt + 3);
    const char* wb = smem + (kt & 1) * BUF + wn0 * ROWB;
    const char* xb = smem + (kt & 1) * BUF + TN * ROWB + wm0 * ROWB;
    char* nbase = smem + ((kt + 1) & 1) * BUF;
    const long long nkoff = (long long)(kt + 1) * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Vec8<T>::type a[FN], b[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) a[i] = lds_read8<T>(wb + i * 16 * ROWB + foff[ks]);
#pragma unroll
      for (int j = 0; j < FM; ++j) b[j] = lds_read8<T>(xb + j * 16 * ROWB + foff[ks]);
#pragma unroll
      for (int i = 0; i < FN; ++i) {
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<T>(a[i], b[j], acc[i][j]);
        if constexpr (SPREAD) {
          // group index 0 .. 2*FN-1; IW + IX loads are spread over the groups, phase-shifted per wave half
          constexpr int NG = 2 * FN, NL = IW + IX;
          static_assert(NG % NL == 0, "loads must divide the MFMA groups");
          constexpr int STEP = NG / NL;
          const int gi = ks * FN + i;
          if (more && (gi % STEP) == (STEP > 1 ? spread_phase : 0)) {
            const int l = gi / STEP;  // compile-time after unrolling
            if (l < IW) glds16(srcW[l] + nkoff, nbase + (wave_q0 + l * NT) * 16);
            else glds16(srcX[l - IW] + nkoff, nbase + TN * ROWB + (wave_q0 + (l - IW) * NT) * 16);
          }
        }
      }
    }
    if (STAMP) {
      tq[4] = stamp();
#pragma unroll
      for (int e = 0; e < 4; ++e) tsum[e] += tq[e + 1] - tq[e];
    }
  }
  const unsigned long long t_loop1 = STAMP ? stamp() : 0;
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, NT, EPI>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
  if (STAMP && args.dbg && lane == 0 && blockIdx.x < 4096) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = stamp();
    unsigned long long* d = args.dbg + ((size_t)blockIdx.x * (WGN * WGM) + wave) * 8;
    d[0] = tsum[0]; d[1] = tsum[1]; d[2] = tsum[2]; d[3] = tsum[3];
    d[4] = t_loop0 - t_entry; d[5] = t_loop1 - t_loop0; d[6] = t_end - t_loop1; d[7] = t_entry;
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
template <typename T, int EPI, bool NODMA = false>  // NODMA: diagnostic only (wrong results): loaders stop after tile 1
__global__ void __launch_bounds__(768) gemm_ws_kernel(const GemmArgs args) {
  constexpr int TN = 256, TM = 256, BK = 64, ROWB = BK * 2;
  constexpr int WGM = 4, WTN = 128, WTM = 64, FN = 8, FM = 4;
  constexpr int BUF = (TN + TM) * ROWB;
  constexpr int NLD = 16;  // LDS-DMA instructions per loader lane per K tile: 4096 chunks / 256 lanes
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
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = lt + (i & 7) * 256;  // chunk inside the operand tile (2048 chunks each)
      const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
      if (i < 8) {
        src[i] = (const char*)P.W + ((long long)(n0 + row) * K + c * 8) * 2;
      } else {
        const int m = min(m0 + row, M - 1);
        src[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
      }
    }
    const int wq0 = (wave - 8) * 64;
    auto stage = [&](int buf, int kt) {
      char* base = smem + buf * BUF;
      const long long koff = (long long)kt * ROWB;
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        glds16(src[i] + koff, base + (i < 8 ? 0 : TN * ROWB) + (wq0 + (i & 7) * 256) * 16);
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 1 < nk && !(NODMA && kt >= 1)) stage((kt + 1) & 1, kt + 1);
    }
    if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) {
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
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, 512, EPI>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
}

// =================================================================================================
// ws2: warp-specialised 256x256, 4-slot LDS ring of 32-deep K slices, NO barrier in the K loop
// =================================================================================================
// With a barrier per K tile every compute wave restarts in lockstep: the two waves of a SIMD read
// their first fragments at the same moment and the matrix pipe idles ~900 of ~3000 cycles per tile
// ("ws without DMA" still stops at 1.25 PF).  Here loaders and compute waves hand slots over through
// monotonic LDS counters instead: full[s] (+1 by the slot's loader wave when its DMA has landed:
// vmcnt(0), then ds_add) and done[s] (+1 per compute wave after its last fragment read of the slot).
// A compute wave waits for full[s] >= use + 1 -- the poll for the next slot is issued early and
// normally already satisfied -- and never meets the other waves, so SIMD partners drift apart and
// cover each other's read bubbles; loader wave s refills slot s once done[s] >= 8 * use, so up to 3
// slots are in flight or landed ahead of the one being consumed.  Every spin is bounded (a hung grid would take the GPU down): on give-up the
// kernel finishes with wrong data and sets args.dbg[0] if a debug buffer is installed.
__device__ __forceinline__ bool spin_ge(volatile unsigned* p, unsigned target) {
  for (int i = 0; i < (1 << 16); ++i) {  // ~5 ms; a real wait is a few microseconds
    if (*p >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

template <typename T, int EPI>
__global__ void __launch_bounds__(768) gemm_ws2_kernel(const GemmArgs args) {
  constexpr int TN = 256, TM = 256, SK = 32, SROW = SK * 2, NS = 4;
  constexpr int WGM = 4, WTN = 128, WTM = 64, FN = 8, FM = 4;
  constexpr int SLOT = (TN + TM) * SROW;  // 32 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile unsigned* full = reinterpret_cast<volatile unsigned*>(smem + NS * SLOT);
  volatile unsigned* done = full + NS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n0, m0;
  const GemmProb& P = pick_tile<TN, TM>(args, n0, m0);
  const int K = P.K, M = P.M;
  const int nslots = K / SK;
  if (tid < 2 * NS) full[tid] = 0;
  __syncthreads();
  bool ok = true;

  if (wave >= 8) {
    // ------------------------------- loader waves -------------------------------
    // loader lw owns ring slot lw: it refills it (one 32 KiB slot = 32 LDS-DMA instructions of this
    // wave) as soon as all 8 compute waves are done with the previous occupant, and signals it the
    // moment the data has landed -- no loader ever waits on another slot's consumers.
    const int lw = wave - 8;
    // chunk q = lane + 64 i of an operand image (1024 chunks): row = (lane >> 2) + 16 i, physical chunk
    // lane & 3, and the swizzle term (-(row >> 2)) & 3 = (-(lane >> 4)) & 3 does not depend on i
    const int row0 = lane >> 2;
    const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const char* srcW = (const char*)P.W + ((long long)(n0 + row0) * K + c * 8) * 2;
    const long long strideW = (long long)16 * K * 2;
    const char* srcX[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = min(m0 + row0 + 16 * i, M - 1);
      srcX[i] = (const char*)P.A + (view_off(P.a, m) + c * 8) * 2;
    }
    for (int t = lw; t < nslots; t += NS) {
      const int use = t / NS;
      if (use > 0 && ok) ok = spin_ge(done + lw, 8u * use);  // after a give-up: run through, never wait again
      asm volatile("" ::: "memory");
      char* base = smem + lw * SLOT;
      const long long koff = (long long)t * SROW;
#pragma unroll
      for (int i = 0; i < 16; ++i) glds16(srcW + i * strideW + koff, base + i * 1024);
#pragma unroll
      for (int i = 0; i < 16; ++i) glds16(srcX[i] + koff, base + TN * SROW + i * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(const_cast<unsigned*>(full) + lw, 1u);
    }
  } else {
    // --------------------------------- compute waves ---------------------------------
    const int wn0 = (wave / WGM) * WTN;
    const int wm0 = (wave % WGM) * WTM;
    const int foff = (lane & 15) * SROW + ((((lane >> 4)) ^ ((-((lane & 15) >> 2)) & 3)) << 4);
    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned seen = full[0];
    for (int t = 0; t < nslots; ++t) {
      const int s = t & (NS - 1), use = t / NS;
      if (seen < (unsigned)(use + 1) && ok) ok = spin_ge(full + s, (unsigned)(use + 1));
      asm volatile("" ::: "memory");
      if (t + 1 < nslots) seen = full[(t + 1) & (NS - 1)];  // early poll for the next slot
      const char* wb = smem + s * SLOT + wn0 * SROW + foff;
      const char* xb = smem + s * SLOT + TN * SROW + wm0 * SROW + foff;
      typename Vec8<T>::type b[FM];
#pragma unroll
      for (int j = 0; j < FM; ++j) b[j] = lds_read8<T>(xb + j * 16 * SROW);
      typename Vec8<T>::type a_cur = lds_read8<T>(wb);
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        typename Vec8<T>::type a_nxt = a_cur;
        if (i + 1 < FN) a_nxt = lds_read8<T>(wb + (i + 1) * 16 * SROW);
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<T>(a_cur, b[j], acc[i][j]);
        a_cur = a_nxt;
      }
      asm volatile("" ::: "memory");  // the slot's reads are all consumed by MFMAs issued above
      if (lane == 0) atomicAdd(const_cast<unsigned*>(done) + s, 1u);
    }
    if (!ok && args.dbg && lane == 0) args.dbg[0] = 1;
    __syncthreads();  // pairs with the loaders' barrier below: every slot read is over
    if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) {
      epilogue_lds16<T, TN, TM, FN, FM, 512, EPI>(P, acc, smem, n0, m0, wn0, wm0, tid);
    } else {
      epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
    }
    return;
  }
  if (!ok && args.dbg && lane == 0) args.dbg[0] = 1;
  __syncthreads();
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) __syncthreads();  // the one inside epilogue_lds16
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
    srcW[i] = (const char*)P.W + ((long long)(n0 + row) * K + c * 8) * 2;
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
  if constexpr (EPI == EPI_OP || EPI == EPI_GELU_OP || EPI == EPI_KV) {
    __syncthreads();
    epilogue_lds16<T, TN, TM, FN, FM, NT, EPI>(P, acc, smem, n0, m0, wn0, wm0, tid);
  } else {
    epilogue<T, FN, FM, EPI>(P, acc, n0 + wn0, m0 + wm0, lane);
  }
}

int g_force_cfg = -1;
int g_variant = 5;  // 5 (default) = warp-specialised 256x256 + two-buffer loop for the small tiles;
                    // 0 = ring, 1 = two-buffer loop, 2 = + L2 prefetch, 3 = + spread DMA issue, 4 = stamped (diagnostic)
unsigned long long* g_dbg = nullptr;

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
    default: return -2;                                     \
  }

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_v1(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 128;
  MRA_EPI_SWITCH((launch_k(gemm_kernel<T, TN, TM, WGN, WGM, E, false>, a, WGN * WGM * 64, lds, stream)))
}

template <typename T>
int launch_ws(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (256 + 256) * 128;
  MRA_EPI_SWITCH((launch_k(gemm_ws_kernel<T, E>, a, 768, lds, stream)))
}

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_k128(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 256;
  MRA_EPI_SWITCH((launch_k(gemm_k128_kernel<T, TN, TM, WGN, WGM, E>, a, WGN * WGM * 64, lds, stream)))
}

template <typename T>
int launch_ws2(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 4 * (256 + 256) * 64 + 64;
  MRA_EPI_SWITCH((launch_k(gemm_ws2_kernel<T, E>, a, 768, lds, stream)))
}

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_v1stamp(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 128;
  if (epi != EPI_KV) return -2;
  return launch_k(gemm_kernel<T, TN, TM, WGN, WGM, EPI_KV, false, false, true>, a, WGN * WGM * 64, lds, stream);
}

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_v1spread(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 128;
  MRA_EPI_SWITCH((launch_k(gemm_kernel<T, TN, TM, WGN, WGM, E, false, true>, a, WGN * WGM * 64, lds, stream)))
}

template <typename T, int TN, int TM, int WGN, int WGM>
int launch_v1pf(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = 2 * (TN + TM) * 128 + WGN * WGM * 512;
  MRA_EPI_SWITCH((launch_k(gemm_kernel<T, TN, TM, WGN, WGM, E, true>, a, WGN * WGM * 64, lds, stream)))
}

template <typename T, int TN, int TM, int WGN, int WGM, int NS, int SPI>
int launch_ring(const GemmArgs& a, int epi, hipStream_t stream) {
  constexpr size_t lds = (size_t)NS * (TN + TM) * 64;
  MRA_EPI_SWITCH((launch_k(gemm_ring_kernel<T, TN, TM, WGN, WGM, NS, SPI, E>, a, WGN * WGM * 64, lds, stream)))
}

template <typename T>
int launch_t(const GemmArgs& a, int cfg, int epi, hipStream_t stream) {
  if (g_variant == 8 && cfg < 2) {  // A/B: 128-deep steps wherever K allows
    bool k128 = true;
    for (int g = 0; g < a.ngroups; ++g) k128 = k128 && a.p[g].K % 128 == 0;
    if (k128) return cfg == 1 ? launch_k128<T, 128, 128, 2, 2>(a, epi, stream) : launch_k128<T, 64, 64, 2, 2>(a, epi, stream);
  }
  if (g_variant == 5 && cfg == 0) {
    // default: 128-deep steps pay on the 64x64 tile once the K loop is long (FFN down-projection,
    // K = 3072: 367 -> 460 TF/s); at K = 768 the launch is prologue/epilogue-bound and nothing changes
    bool k128 = true;
    for (int g = 0; g < a.ngroups; ++g) k128 = k128 && a.p[g].K % 128 == 0 && a.p[g].K >= 2048;
    if (k128) return launch_k128<T, 64, 64, 2, 2>(a, epi, stream);
  }
  if (g_variant == 7 && cfg == 2) return launch_ws2<T>(a, epi, stream);
  if ((g_variant == 5 || g_variant == 8) && cfg == 2) return launch_ws<T>(a, epi, stream);
  if (g_variant == 6 && cfg == 2 && epi == EPI_KV) return launch_k(gemm_ws_kernel<T, EPI_KV, true>, a, 768, 2 * 512 * 128, stream);
  if (g_variant == 4) return launch_v1stamp<T, 256, 256, 2, 4>(a, epi, stream);
  if (g_variant == 3) {
    if (cfg == 2) return launch_v1spread<T, 256, 256, 2, 4>(a, epi, stream);
    if (cfg == 1) return launch_v1spread<T, 128, 128, 2, 2>(a, epi, stream);
    return launch_v1spread<T, 64, 64, 2, 2>(a, epi, stream);
  }
  if (g_variant == 2) {
    if (cfg == 2) return launch_v1pf<T, 256, 256, 2, 4>(a, epi, stream);
    if (cfg == 1) return launch_v1pf<T, 128, 128, 2, 2>(a, epi, stream);
    return launch_v1pf<T, 64, 64, 2, 2>(a, epi, stream);
  }
  if (g_variant == 1 || g_variant == 5 || g_variant == 7 || g_variant == 8) {
    if (cfg == 2) return launch_v1<T, 256, 256, 2, 4>(a, epi, stream);
    if (cfg == 1) return launch_v1<T, 128, 128, 2, 2>(a, epi, stream);
    return launch_v1<T, 64, 64, 2, 2>(a, epi, stream);
  }
  // ring: 256x256 -> 5 slots x 32 KiB (all of LDS), 128x128 -> 6 x 16 KiB, 64x64 -> 8 x 8 KiB; 2 slots per barrier
  if (cfg == 2) return launch_ring<T, 256, 256, 2, 4, 5, 2>(a, epi, stream);
  if (cfg == 1) return launch_ring<T, 128, 128, 2, 2, 6, 2>(a, epi, stream);
  return launch_ring<T, 64, 64, 2, 2, 8, 2>(a, epi, stream);
}

constexpr int kTile[3] = {64, 128, 256};

}  // namespace

void gemm_force_config(int cfg) { g_force_cfg = cfg; }
void gemm_force_variant(int v) { g_variant = v; }
void gemm_set_debug_buffer(unsigned long long* p) { g_dbg = p; }

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
    if (p.K % 64 || p.N % t) return -1;
    if (p.a.rpi <= 0 || (epi != EPI_KV && p.c.rpi <= 0)) return -1;
    if (epi == EPI_RES_F32 && (!p.R || p.r.rpi <= 0)) return -1;
    if (epi == EPI_KV && (p.kv_tokens <= 0 || p.kv_heads <= 0 || p.kv_items <= 0)) return -1;
    if ((epi == EPI_OP || epi == EPI_GELU_OP) && ((p.c.ld & 7) || (p.c.item_stride & 7))) return -1;  // 16-byte stores
    p.mtiles = (p.M + t - 1) / t;
    p.ntiles = p.N / t;
    p.tile_begin = tiles;
    tiles += p.mtiles * p.ntiles;
  }
  if (ngroups == 1) a.p[1] = a.p[0];
  a.total_tiles = tiles;
  a.dbg = g_dbg;
  return op_dtype == OP_F16 ? launch_t<f16>(a, cfg, epi, stream) : launch_t<bf16>(a, cfg, epi, stream);
}

}  // namespace mra
