// The two big products of the folded cross-attention (mra_abi.hip, reference arithmetic HF modeling_instructblip.py:464-515
// re-associated), as a streaming kernel built for what bounds them on MI355X: bytes in flight per CU.
//
//   scores  P~[r][k] = exp2(alpha * (Q'[r] . enc[k]) - m_tile[r])      [384 (head, query) rows] x [kv tokens], K = E
//   pv      U[r][e]  = (1 / L[r]) * sum_k g[r][tile(k)] P~[r][k] enc[k][e]                       [384] x [E],   K = kv
//
// Both are [384 x K] x [K x 176]-shaped tiles whose 176-wide operand (a slab of the item's encoder tokens) comes from
// HBM exactly once; the loader-wave GEMM this replaces (gemm.hip, 176 x 384 tile, two 64-deep LDS stages) kept ONE K step
// of that stream in flight and spent 4.6 k cycles per step against 2.1 k of MFMA work.  Here:
//   * 8 waves, no loader waves (2 waves per SIMD -> 256 VGPRs).  Each wave owns 48 of the 384 rows and takes its row
//     operand (Q' / P~: never shared between waves) STRAIGHT to registers in MFMA fragment layout, NSETS - 1 K steps ahead
//     (rotating register sets) -- it no longer crosses LDS at all;
//   * the LDS then belongs to the shared slab alone: a ring of 4 slots of one 64-deep K step each (LDS-DMA, non-temporal),
//     three steps ahead at issue time;
//   * one raw s_barrier per K step; every vector-memory operation of the loop is inline assembly so the counted
//     s_waitcnt vmcnt is exact (the compiler neither sees nor waits for any of them);
//   * the softmax is split over the 176-column tiles as before (tile maximum / tile sum in the scores epilogue), but the
//     tile maximum is rounded UP to an integer: the per-tile factor g = 2^(m_tile - m_row) is then a power of two, exact
//     in f16, and is applied to the P~ fragments in registers (v_pk_mul_f16) on their way into the MFMA; 1/L goes into
//     the pv epilogue.  P~ is written once and read once: the in-place rescale pass (0.4 GB, 66 us per layer) is gone,
//     and P~ keeps entries down to 6e-8 of its TILE maximum instead of 6e-8 of the row maximum.
// The row operands live in HBM in a BLOCKED layout, [row block of 16][chunk of 8 k][16 rows][8 k]: a fragment load (lane l:
// row l & 15, chunk l >> 4) then reads 1 KiB contiguous per wave instruction instead of 64 sixteen-byte pieces of 16
// different rows (first cut, row-major: the vector L1 looked up 64 tags per load and both products ran 25 % SLOWER than
// the loader-wave GEMMs), and the scores epilogue writes whole 256-byte blocks.  P~ is internal; Q' is re-packed by a
// small kernel (fold_pack_kernel) after the per-head Q' GEMM.
// The in-flight register sets are ordinary asm outputs: the build must show NO scratch use and no compiler copy of them
// between load and wait (cdna_hip_programming.md 5.7 item 1); `make` fails otherwise (Makefile: fold_stream audit).
// f16 operands only (the bf16 configuration keeps the loader-wave path).
#include <type_traits>

#include "kernels.h"
#include "mra_common.h"

namespace mra {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 384;        // (head, query) rows of an item = 8 waves x 48
constexpr int TN = 176;          // slab width (tokens for scores, encoder channels for pv)
constexpr int BK = 64;
constexpr int FN = TN / 16, FM = 3;
constexpr int SLOT = 24 * 1024;  // 22 DMA pieces of 1 KiB + 2 dummy pieces (every wave issues exactly three per step)
constexpr int NSLOT = 4;
constexpr int GT_OFF = NSLOT * SLOT;   // pv: table of tile factors [384][stat_ld] f16 behind the ring

struct FoldArgs {
  const f16* A;        // scores: Q' [items][384][lda];  pv: P~ [items][384][lda]
  const f16* W;        // scores: enc [items][kv][ldw] (row n = token, K = channels);  pv: the same tokens read K-major
  f16* C;              // scores: P~ [items][384][ldc];  pv: U [items][384][ldc]
  long long a_bs, w_bs, c_bs;   // elements per item
  int lda, ldw, ldc;
  int K;               // scores: E;  pv: kvp (multiple of 128)
  int n_valid;         // scores: kv (tokens);  pv: E
  int k_rows;          // pv: kv (valid K rows of W; rows past it are clamped and meet P~ = 0)
  int ntiles;          // 176-wide tiles along N
  int items;
  float alpha;         // scores: scale in log2 units
  float* stat_m;       // scores out: [items * 384][stat_ld] integer tile maxima (log2 units)
  float* stat_l;       //             tile sums of the f16 P~ values
  const f16* gexp;     // pv in: [items * 384][stat_ld] = 2^(m_tile - m_row)
  const float* ginv;   // pv in: [items * 384] = 1 / L
  int stat_ld;
  int zero_from, zero_to;   // scores: the last tile also zeroes P~ columns [zero_from, zero_to)
};

// LDS-DMA, 16 bytes per lane, non-temporal, through a buffer descriptor: source = descriptor base + scalar offset + per-lane
// byte offset, destination = m0 + lane * 16.  Bytes past the descriptor's size read as ZERO (hardware range check): the
// encoder-token rows past the last valid one need no clamping -- they meet P~ = 0 (pv) or masked score columns (scores).
typedef int i32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_nt(i32x4s rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gload16(i32x4& dst, const void* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
struct ASet { i32x4 f[FM][2]; };
template <int N>
__device__ __forceinline__ void wait_set(ASet& s) {   // all but the N youngest vector-memory operations are done
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(s.f[0][0]), "+v"(s.f[0][1]), "+v"(s.f[1][0]), "+v"(s.f[1][1]), "+v"(s.f[2][0]), "+v"(s.f[2][1]) : "n"(N) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const void* p) {   // make wave-uniformity provable ("s" operands)
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// MODE 0 = scores, MODE 1 = pv.  NSETS register sets of the row operand: stage kt + NSETS - 1 is issued in step kt.
template <int MODE, int NSETS>
__global__ void __launch_bounds__(512, 2) fold_stream_kernel(const FoldArgs P) {
  static_assert(NSETS == 2 || NSETS == 3, "two or three register sets");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  int id = blockIdx.x;
  {   // XCD-contiguous remap (bijective for any grid): the tiles of one item run on one XCD and share its L2
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int item = id / P.ntiles, tile = id - item * P.ntiles;
  const int n0 = tile * TN;
  const int nk = P.K / BK;

  // ---- the row operand: lane l of fragment j holds row 48 wave + 16 j + (l & 15), k = 64 kt + 32 ks + 8 (l >> 4) .. + 7.
  // Addresses are (scalar base of the item + K-step) + a per-lane byte offset that never changes: 3 VGPRs for the loop.
  const int row0 = wave * 48 + (lane & 15);
  const char* a_item = uniform_ptr(P.A + (long long)item * P.a_bs);
  // blocked layout: row block rb = 3 wave + j, chunk c: 16-byte piece at ((rb * (K / 8) + c) * 16 + (lane & 15)) * 16 bytes
  unsigned aoff[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) aoff[j] = (unsigned)((((3 * wave + j) * (P.K / 8) + (lane >> 4)) * 16 + (lane & 15)) * 16);
  auto load_a = [&](ASet& s, int kt) __attribute__((always_inline)) {
    const char* base = a_item + (long long)min(kt, nk - 1) * (BK / 8 * 256);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      gload16<0>(s.f[j][0], base, aoff[j]);
      gload16<1024>(s.f[j][1], base, aoff[j]);
    }
  };

  // ---- the slab: three 1-KiB pieces per wave and K step (pieces wave, wave + 8, wave + 16; pieces 22, 23 are dummies
  // that repeat piece 21 into the pad behind the slot so that every wave issues exactly three)
  const char* w_item = uniform_ptr(P.W + (long long)item * P.w_bs);
  const unsigned w_bytes = (unsigned)((long long)P.k_rows * P.ldw * 2);                 // the item's tokens; beyond: zeros
  i32x4s w_rsrc;   // raw buffer descriptor: base, stride 0, size in bytes, data format word (cdna_hip_programming.md T8)
  w_rsrc[0] = (int)(unsigned)(unsigned long long)w_item;
  w_rsrc[1] = (int)((unsigned)((unsigned long long)w_item >> 32) & 0xffffu);
  w_rsrc[2] = (int)__builtin_amdgcn_readfirstlane(w_bytes);
  w_rsrc[3] = 0x00020000;
  unsigned woff[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int q = min(wave + 8 * u, 21) * 64 + lane;    // 16-byte chunk of the K-step tile
    if (MODE == 0) {   // [176 rows][64 k], 128-byte rows, chunk index XOR-swizzled on the SOURCE (the DMA writes linearly)
      const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
      woff[u] = (unsigned)(((n0 + row) * P.ldw + c * 8) * 2);
    } else {           // [64 k][176 n], 352-byte rows, linear (fragments come from transposed reads)
      const int kr = q / (TN / 8);
      woff[u] = (unsigned)((kr * P.ldw + n0 + (q - kr * (TN / 8)) * 8) * 2);
    }
  }
  auto load_w = [&](int kt) __attribute__((always_inline)) {
    const int kc = min(kt, nk - 1);
    const unsigned slot = lds0 + (unsigned)(kt % NSLOT) * SLOT;
    const unsigned soff = MODE == 0 ? (unsigned)kc * (BK * 2) : (unsigned)kc * BK * P.ldw * 2;
#pragma unroll
    for (int u = 0; u < 3; ++u) dma16_nt(w_rsrc, woff[u], soff, slot + (unsigned)(wave + 8 * u) * 1024);
  };

  // fragment read addresses inside a slot
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int r = lane & 15;
    foff[ks] = r * 128 + (((4 * ks + (lane >> 4)) ^ ((r >> 1) & 7)) << 4);
  }
  const int wtr = (8 * (lane >> 4) + ((lane & 15) >> 2)) * (TN * 2) + (lane & 3) * 8;   // pv: row 8 g + (j >> 2), 4 columns from 4 (j & 3)

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue.  The issue order fixes the vmcnt arithmetic: step kt issues [6 row loads of stage kt + NSETS - 1]
  // [3 slab pieces of stage kt + 3]; at the top of step kt everything up to the row loads of stage kt must be complete,
  // i.e. all but the youngest 3 + 9 (NSETS - 2) operations.
  ASet s0, s1, s2;
  load_w(0);
  load_a(s0, 0);
  load_w(1);
  if (NSETS == 3) load_a(s1, 1);
  load_w(2);
  constexpr int INFLIGHT = 3 + 9 * (NSETS - 2);
  if (MODE == 1) {   // tile factors of this item's 384 rows -> LDS (plain loads: waited by the compiler before the writes)
    const int n16 = ROWS * P.stat_ld / 8;
    const i32x4* src = (const i32x4*)(P.gexp + (long long)item * ROWS * P.stat_ld);
    for (int c = tid; c < n16; c += 512) *(i32x4*)(smem + GT_OFF + c * 16) = src[c];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  auto step = [&](ASet& cur, ASet& nxt, int kt) __attribute__((always_inline)) {
    wait_set<INFLIGHT>(cur);             // stage kt of the rows and (older) this wave's pieces of slab stage kt have landed
    __builtin_amdgcn_s_barrier();        // ... for every wave; and every wave is done reading slab stage kt - 1
    asm volatile("" ::: "memory");
    load_a(nxt, kt + NSETS - 1);         // into the register set consumed one step ago
    load_w(kt + 3);                      // into the slot read one step ago
    const char* slot = smem + (kt % NSLOT) * SLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<f16>::type b[FM];
#pragma unroll
      for (int j = 0; j < FM; ++j) b[j] = __builtin_bit_cast(Vec8<f16>::type, cur.f[j][ks]);
      if (MODE == 1) {
        // P~ -> g P~, in place (v_pk_mul_f16, factor broadcast from the low half): the factor of the 176-column score tile
        // this lane's 8 consecutive k fall in (176 = 22 x 8)
        const int t = (kt * BK + 32 * ks + 8 * (lane >> 4)) / TN;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const unsigned g = *(const unsigned short*)(smem + GT_OFF + ((row0 + 16 * j) * P.stat_ld + t) * 2);
          asm volatile("v_pk_mul_f16 %0, %0, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f16 %1, %1, %4 op_sel_hi:[1,0]\n\t"
                       "v_pk_mul_f16 %2, %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f16 %3, %3, %4 op_sel_hi:[1,0]\n\ts_nop 1"
                       : "+v"(cur.f[j][ks][0]), "+v"(cur.f[j][ks][1]), "+v"(cur.f[j][ks][2]), "+v"(cur.f[j][ks][3]) : "v"(g));
          b[j] = __builtin_bit_cast(Vec8<f16>::type, cur.f[j][ks]);
        }
        // fragment of slab tile i: lane (column n = 16 i + (lane & 15)) needs k = 32 ks + 8 (lane >> 4) .. + 7 = two transposed
        // 4 x 16 blocks of the [k][n] tile (rows 8 g + 0..3 and + 4..7)
        constexpr int WP = TN * 2;
        const char* tb = slot + wtr + ks * 32 * WP;
        i16x4 c0 = lds_read_tr4(tb), c1 = lds_read_tr4(tb + 4 * WP);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          i16x4 n0_ = c0, n1_ = c1;
          if (i + 1 < FN) { n0_ = lds_read_tr4(tb + (i + 1) * 32); n1_ = lds_read_tr4(tb + (i + 1) * 32 + 4 * WP); }
          i16x8 v;
          v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3]; v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
          const Vec8<f16>::type a_cur = __builtin_bit_cast(Vec8<f16>::type, v);
#pragma unroll
          for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<f16>(a_cur, b[j], acc[i][j]);
          c0 = n0_; c1 = n1_;
          // keep the reads one fragment ahead of the MFMAs and no further: the scheduler otherwise hoists the whole K step's
          // 44 reads above the MFMAs and the register allocator spills
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
      } else {
        Vec8<f16>::type a_cur = lds_read8<f16>(slot + foff[ks]);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          Vec8<f16>::type a_nxt = a_cur;
          if (i + 1 < FN) a_nxt = lds_read8<f16>(slot + (i + 1) * 16 * 128 + foff[ks]);
#pragma unroll
          for (int j = 0; j < FM; ++j) acc[i][j] = mfma16<f16>(a_cur, b[j], acc[i][j]);
          a_cur = a_nxt;
        }
      }
    }
  };

  int kt = 0;
  if (NSETS == 3) {
    for (; kt + 3 <= nk; kt += 3) {
      step(s0, s2, kt);
      step(s1, s0, kt + 1);
      step(s2, s1, kt + 2);
    }
    if (kt < nk) step(s0, s2, kt++);
    if (kt < nk) step(s1, s0, kt++);
  } else {
    for (; kt + 2 <= nk; kt += 2) {
      step(s0, s1, kt);
      step(s1, s0, kt + 1);
    }
    if (kt < nk) step(s0, s1, kt++);
  }
  // the clamped look-ahead loads of the last steps are still in flight (registers that are never read, LDS slots nobody
  // reads any more): drain them before the wave reuses the registers or exits
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0.f[0][0]), "+v"(s0.f[0][1]), "+v"(s0.f[1][0]), "+v"(s0.f[1][1]), "+v"(s0.f[2][0]), "+v"(s0.f[2][1]),
                                      "+v"(s1.f[0][0]), "+v"(s1.f[0][1]), "+v"(s1.f[1][0]), "+v"(s1.f[1][1]), "+v"(s1.f[2][0]), "+v"(s1.f[2][1])::"memory");
  if (NSETS == 3)
    asm volatile("" : "+v"(s2.f[0][0]), "+v"(s2.f[0][1]), "+v"(s2.f[1][0]), "+v"(s2.f[1][1]), "+v"(s2.f[2][0]), "+v"(s2.f[2][1])::"memory");

  // ---- epilogues: lane owns C[m][n .. n + 3], m = row0 + 16 j, n = n0 + 16 i + 4 (lane >> 4)
  const int ln = (lane >> 4) * 4;
  f16* Cb = P.C + (long long)item * P.c_bs;
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = row0 + 16 * j;
      float mx = -3.0e38f;
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[i][j][e] *= P.alpha;
          if (n0 + i * 16 + ln + e < P.n_valid) mx = fmaxf(mx, acc[i][j][e]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mx = ceilf(mx);                       // integer reference: the tile factor 2^(m_tile - m_row) is exact in f16
      float l = 0.f;
      // P~ in the blocked layout of the pv product's row operand: [row block][chunk of 8 k][16 rows][8 k]
      f16* cblk = Cb + ((long long)(3 * wave + j) * (P.ldc / 8) * 16 + (lane & 15)) * 8 + 4 * ((lane >> 4) & 1);
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int n = n0 + i * 16 + ln;
        Vec4<f16>::type o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = n + e < P.n_valid ? __builtin_amdgcn_exp2f(acc[i][j][e] - mx) : 0.f;
          o[e] = (f16)p;
          l += (float)o[e];                 // the sum of what the pv product will actually read
        }
        *reinterpret_cast<Vec4<f16>::type*>(cblk + (long long)(n >> 3) * 128) = o;
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      if (ln == 0) {
        const long long si = ((long long)item * ROWS + m) * P.stat_ld + tile;
        P.stat_m[si] = mx;
        P.stat_l[si] = l;
      }
      if (tile == P.ntiles - 1) {           // the pv product's K runs to the padded row length: those columns must be zero
        const Vec4<f16>::type z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        for (int n = P.zero_from + ln; n < P.zero_to; n += 16) *reinterpret_cast<Vec4<f16>::type*>(cblk + (long long)(n >> 3) * 128) = z;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = row0 + 16 * j;
      const float inv = P.ginv[(long long)item * ROWS + m];
      f16* crow = Cb + (long long)m * P.ldc;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        Vec4<f16>::type o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[i][j][e] * inv);
        *reinterpret_cast<Vec4<f16>::type*>(crow + n0 + i * 16 + ln) = o;
      }
    }
  }
}

// Row-major [rows][K] -> the blocked layout [rows / 16][K / 8][16][8] (rows % 16 == 0, K % 8 == 0).  One thread per 16-byte
// piece, indexed by its DESTINATION so that stores are contiguous; the loads of a 16-lane group walk 16 rows.
__global__ void __launch_bounds__(256) fold_pack_kernel(const i32x4* src, i32x4* dst, long long pieces, int kc) {
  const long long d = (long long)blockIdx.x * 256 + threadIdx.x;
  if (d >= pieces) return;
  const int m = (int)(d & 15);
  const long long blk = d >> 4;            // rb * kc + c
  const long long rb = blk / kc;
  const int c = (int)(blk - rb * kc);
  dst[d] = src[(rb * 16 + m) * kc + c];
}

// One thread per (item, row): m_row = max m_tile, g_tile = 2^(m_tile - m_row) rounded to f16 (exact, or 0 below 2^-24),
// L = sum g_tile l_tile with the factors the pv product will really use.
__global__ void __launch_bounds__(256) fold_stats_kernel(const float* stat_m, const float* stat_l, int rows, int ntiles, int stat_ld, f16* gexp,
                                                         float* ginv) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float* sm = stat_m + (long long)row * stat_ld;
  const float* sl = stat_l + (long long)row * stat_ld;
  f16* g = gexp + (long long)row * stat_ld;
  float m = -3.0e38f;
  for (int t = 0; t < ntiles; ++t) m = fmaxf(m, sm[t]);
  float L = 0.f;
  for (int t = 0; t < ntiles; ++t) {
    const f16 gt = (f16)__builtin_amdgcn_exp2f(sm[t] - m);
    g[t] = gt;
    L += (float)gt * sl[t];
  }
  for (int t = ntiles; t < stat_ld; ++t) g[t] = (f16)0.f;
  ginv[row] = 1.0f / L;
}

#ifndef MRA_FOLD_SETS_SCORES
#define MRA_FOLD_SETS_SCORES 3
#endif
#ifndef MRA_FOLD_SETS_PV
#define MRA_FOLD_SETS_PV 2
#endif

}  // namespace

int fold_stream_stat_ld(int kvp) { return ((kvp + TN - 1) / TN + 7) / 8 * 8; }

bool fold_stream_supported(int rows, int E, int kv, int kvp, int op_dtype) {
  if (op_dtype != OP_F16 || rows != ROWS || E % TN || E % BK || kvp % 128 || kv < 1 || kvp < kv) return false;
  if ((long long)kvp * ROWS * 2 >= (1ll << 32) || (long long)kv * E * 2 >= (1ll << 32)) return false;   // 32-bit per-lane byte offsets
  return GT_OFF + ROWS * fold_stream_stat_ld(kvp) * 2 <= 160 * 1024;
}

int launch_fold_stream(const FoldStreamArgs& a, hipStream_t stream) {
  if (!fold_stream_supported(ROWS, a.E, a.kv, a.kvp, OP_F16) || a.items <= 0) return -1;
  static unsigned long long attr_done = 0;   // bit per device: hipFuncSetAttribute is not a stream operation, do it once
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int stat_ld = fold_stream_stat_ld(a.kvp);
  const int ntiles_s = (a.kv + TN - 1) / TN;
  const int lds_s = NSLOT * SLOT, lds_p = GT_OFF + ROWS * stat_ld * 2;
  auto k_scores = fold_stream_kernel<0, MRA_FOLD_SETS_SCORES>;
  auto k_pv = fold_stream_kernel<1, MRA_FOLD_SETS_PV>;
  if (!(attr_done >> (dev & 63) & 1)) {
    if (hipFuncSetAttribute((const void*)k_scores, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s) != hipSuccess) return -3;
    if (hipFuncSetAttribute((const void*)k_pv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -3;
    attr_done |= 1ull << (dev & 63);
  }
  if (a.phase & 1) {
    const long long pieces = (long long)a.items * ROWS * (a.E / 8);
    hipLaunchKernelGGL(fold_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, (const i32x4*)a.qp, (i32x4*)a.qpb, pieces, a.E / 8);
    FoldArgs s{};
    s.A = (const f16*)a.qpb; s.W = (const f16*)a.enc; s.C = (f16*)a.p;
    s.a_bs = (long long)ROWS * a.E; s.w_bs = (long long)a.kv * a.E; s.c_bs = (long long)ROWS * a.kvp;
    s.lda = a.E; s.ldw = a.E; s.ldc = a.kvp;
    s.K = a.E; s.n_valid = a.kv; s.k_rows = a.kv; s.ntiles = ntiles_s; s.items = a.items;
    s.alpha = a.alpha; s.stat_m = a.stat_m; s.stat_l = a.stat_l; s.stat_ld = stat_ld;
    s.zero_from = ntiles_s * TN; s.zero_to = a.kvp;
    hipLaunchKernelGGL(k_scores, dim3(a.items * ntiles_s), dim3(512), lds_s, stream, s);
    hipLaunchKernelGGL(fold_stats_kernel, dim3((a.items * ROWS + 255) / 256), dim3(256), 0, stream, a.stat_m, a.stat_l, a.items * ROWS, ntiles_s,
                       stat_ld, (f16*)a.gexp, a.ginv);
  }
  if (a.phase & 2) {
    FoldArgs p{};
    p.A = (const f16*)a.p; p.W = (const f16*)a.enc; p.C = (f16*)a.u;
    p.a_bs = (long long)ROWS * a.kvp; p.w_bs = (long long)a.kv * a.E; p.c_bs = (long long)ROWS * a.E;
    p.lda = a.kvp; p.ldw = a.E; p.ldc = a.E;
    p.K = a.kvp; p.n_valid = a.E; p.k_rows = a.kv; p.ntiles = a.E / TN; p.items = a.items;
    p.gexp = (const f16*)a.gexp; p.ginv = a.ginv; p.stat_ld = stat_ld;
    hipLaunchKernelGGL(k_pv, dim3(a.items * p.ntiles), dim3(512), lds_p, stream, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace mra
