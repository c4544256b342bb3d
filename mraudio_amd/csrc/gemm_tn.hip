// Weight-gradient GEMM for the Q-Former backward (BASELINE config 5; not in the reference, whose
// Q-Formers are frozen, models/xinstructblip.py:196-204):
//     dW[n][k] (+)= sum_m dY[m][n] * X[m][k]            (nn.Linear: Y = X W^T  =>  dW = dY^T X)
// plus the bias gradient db[n] = sum_m dY[m][n] (colsum_kernel).
//
// Both operands are row-major over the contraction index m, the wrong way round for an MFMA operand
// (a lane wants 8 consecutive m of one column).  Tiles [32 m][64 cols] are staged by LDS-DMA exactly
// like the V tile of the attention kernel (128-byte rows, 16-byte chunk index XOR ((m >> 1) & 1) << 2 on
// the source address) and both fragments come from ds_read_b64_tr_b16 transposed reads:
//     D[32 n x 32 k] += A[32 n x 16 m] * B[16 m x 32 k],   v_mfma_f32_32x32x16, fp32 accumulate.
// One workgroup = 64 n x 64 k (4 waves, one 32x32 tile each) over a three-deep ring of m tiles (two tiles
// in flight behind the one being read, counted vmcnt, one barrier per step).  Gradient GEMMs are small
// here (M = items * S ~ 1 k rows, 144 output tiles for a 768 x 768 weight): the launcher splits M over
// blockIdx.z until the grid has ~512 workgroups and the partial tiles are added with fp32 atomics
// (summation order across splits is not fixed; the gradient buffer is an accumulator anyway).
#include "kernels.h"
#include "mra_common.h"

#include <algorithm>

namespace mra {

namespace {

constexpr int BMT = 32;                 // contraction rows per step
constexpr int TILE_B = BMT * 128;       // bytes per operand tile

__device__ __forceinline__ long long tn_row_off(const RowView& v, int m) {
  const int item = m / v.rpi;
  return (long long)item * v.item_stride + (long long)(m - item * v.rpi) * v.ld;
}

template <typename T>
__device__ __forceinline__ void gemm_tn_body(const GemmTnArgs& a, int nb, int kb);

template <typename T>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const GemmTnArgs a) {
  gemm_tn_body<T>(a, blockIdx.x, blockIdx.y);
}

// several weight gradients in one launch: grid.x runs over the 64-row blocks of every job's dW one after the other
template <typename T>
__global__ void __launch_bounds__(256) gemm_tn_group_kernel(const GemmTnGroup g) {
  int job = 0;
#pragma unroll
  for (int i = 1; i < GEMM_TN_MAX_JOBS; ++i)
    if (i < g.njobs && (int)blockIdx.x >= g.nb_begin[i]) job = i;
  const GemmTnArgs& a = g.j[job];
  if ((int)blockIdx.y * 64 >= a.K) return;          // grid.y covers the widest job (workgroup-uniform)
  gemm_tn_body<T>(a, blockIdx.x - g.nb_begin[job], blockIdx.y);
}

template <typename T>
__device__ __forceinline__ void gemm_tn_body(const GemmTnArgs& a, int nb, int kb) {   // nb, kb: 64-column blocks of dY and of X
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* Y = (const T*)a.dY + (long long)nb * a.y_block_stride;
  const T* X = (const T*)a.X + (long long)kb * a.x_block_stride;
  const int M = a.M;

  // LDS-DMA: 256 chunks of 16 B per operand tile, one per thread: row = tid >> 3, physical chunk tid & 7
  const int srow = tid >> 3, sc = tid & 7;
  const int src_chunk = sc ^ (((srow >> 1) & 1) << 2);
  auto issue = [&](int buf, int m0) {
    const int m = min(m0 + srow, M - 1);  // rows past M are masked below by zeroing their contribution
    char* yb = smem + buf * 2 * TILE_B;
    glds16((const char*)(Y + tn_row_off(a.yv, m)) + src_chunk * 16, yb + wave * 1024);
    glds16((const char*)(X + tn_row_off(a.xv, m)) + src_chunk * 16, yb + TILE_B + wave * 1024);
  };
  // transposed-read lane offsets (same derivation as the V operand of attention.hip): 16-lane group g,
  // lane-in-group i = 4 q4 + p reads 8 bytes of row 4 h + q4 (+ 8 e + 16 s) at column 32 ct + 16 (g & 1) + 4 p
  unsigned lane_off[2];
  {
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, p = i & 3, h = lane >> 5;
    const int row = 4 * h + q4;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int c = 4 * ct + 2 * (g & 1) + (p >> 1);
      const int pc = c ^ (((q4 >> 1) & 1) << 2);
      lane_off[ct] = row * 128 + pc * 16 + (p & 1) * 8;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int ct_n = wave >> 1, ct_k = wave & 1;  // this wave's 32-column halves of the n and k blocks

  f32x16 acc, accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accb[i] = 0.f; }
  const bool do_bias = a.db != nullptr && kb == 0 && ct_k == 0;   // wave-uniform
  typename Vec8<T>::type ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = from_f32<T>(1.0f);
  const int nsteps = (M + BMT - 1) / BMT;
  const int per = (nsteps + (int)gridDim.z - 1) / (int)gridDim.z;
  const int s0 = blockIdx.z * per, s1 = min(nsteps, s0 + per);
  if (s0 >= s1) return;                      // workgroup-uniform
  issue(0, s0 * BMT);
  if (s0 + 1 < s1) issue(1, (s0 + 1) * BMT);
  int buf = 0;
  for (int st = s0; st < s1; ++st) {
    if (st + 1 < s1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st + 2 < s1) issue(buf >= 1 ? buf - 1 : 2, (st + 2) * BMT);   // the tile read one step ago
    const unsigned ybase = lds0 + buf * 2 * TILE_B + lane_off[ct_n];
    buf = buf == 2 ? 0 : buf + 1;
    const unsigned xbase = ybase - lane_off[ct_n] + TILE_B + lane_off[ct_k];
    i16x4 ya0, ya1, ya2, ya3, xb0, xb1, xb2, xb3;
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
        : "=&v"(ya0), "=&v"(ya1), "=&v"(ya2), "=&v"(ya3), "=&v"(xb0), "=&v"(xb1), "=&v"(xb2), "=&v"(xb3)
        : "v"(ybase), "v"(xbase)
        : "memory");
    auto cat = [](i16x4 lo, i16x4 hi) {
      i16x8 v8;
      v8[0] = lo[0]; v8[1] = lo[1]; v8[2] = lo[2]; v8[3] = lo[3];
      v8[4] = hi[0]; v8[5] = hi[1]; v8[6] = hi[2]; v8[7] = hi[3];
      return v8;
    };
    i16x8 A0 = cat(ya0, ya1), A1 = cat(ya2, ya3), B0 = cat(xb0, xb1), B1 = cat(xb2, xb3);
    // rows past M: the clamped source row was loaded again; zero the A elements of those rows.
    // Element j of the fragment of k-step s is row 16 s + 8 (j >> 2) + 4 h + (j & 3).
    const int mrem = M - st * BMT;
    if (mrem < BMT) {
      const int h = lane >> 5;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (8 * (j >> 2) + 4 * h + (j & 3) >= mrem) A0[j] = 0;
        if (16 + 8 * (j >> 2) + 4 * h + (j & 3) >= mrem) A1[j] = 0;
      }
    }
    acc = mfma32<T>(__builtin_bit_cast(typename Vec8<T>::type, A0), __builtin_bit_cast(typename Vec8<T>::type, B0), acc);
    acc = mfma32<T>(__builtin_bit_cast(typename Vec8<T>::type, A1), __builtin_bit_cast(typename Vec8<T>::type, B1), acc);
    if (do_bias) {   // every column of this product is the column sum of the dY tile
      accb = mfma32<T>(__builtin_bit_cast(typename Vec8<T>::type, A0), ones, accb);
      accb = mfma32<T>(__builtin_bit_cast(typename Vec8<T>::type, A1), ones, accb);
    }
  }
  if (do_bias && (lane & 31) == 0) {
    float* pb = a.db + nb * 64 + ct_n * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) unsafeAtomicAdd(pb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), accb[r]);
  }
  // D: column k = lane & 31, row n = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int kcol = kb * 64 + ct_k * 32 + (lane & 31);
  float* out = a.dW + (long long)(nb * 64 + ct_n * 32) * a.ldw + kcol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float* p = out + (long long)n * a.ldw;
    if (gridDim.z > 1) unsafeAtomicAdd(p, acc[r]);
    else *p = a.accumulate ? *p + acc[r] : acc[r];
  }
}

// db[n] (+)= sum_m dY[m][n].  Grid (N / 64, row chunks): a thread reads 16 bytes (8 columns) of one row, 32
// rows per pass; the 32 row partials meet in LDS and one fp32 atomic per column and workgroup lands in db.
constexpr int COLSUM_ROWS = 256;
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const void* dY_, long long block_stride, RowView yv, int M, float* db,
                                                     int accumulate) {
  __shared__ float part[32][65];
  const int tid = threadIdx.x, c8 = tid & 7, r = tid >> 3;
  const T* Y = (const T*)dY_ + (long long)blockIdx.x * block_stride + c8 * 8;
  const int m0 = blockIdx.y * COLSUM_ROWS, m1 = min(M, m0 + COLSUM_ROWS);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll 4
  for (int m = m0 + r; m < m1; m += 32) {
    const typename Vec8<T>::type v = *(const typename Vec8<T>::type*)(Y + tn_row_off(yv, m));
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[r][c8 * 8 + j] = s[j];
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += part[i][tid];
    float* p = db + blockIdx.x * 64 + tid;
    if (gridDim.y > 1) unsafeAtomicAdd(p, t);
    else *p = accumulate ? *p + t : t;
  }
}

}  // namespace

int launch_gemm_tn(const GemmTnArgs& a, int op_dtype, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return -1;
  if (a.N % 64 || a.K % 64) return -1;
  if (a.yv.rpi <= 0 || a.xv.rpi <= 0 || (a.yv.ld & 7) || (a.xv.ld & 7)) return -1;
  const int tiles = (a.N / 64) * (a.K / 64), nsteps = (a.M + BMT - 1) / BMT;
  int splits = (512 + tiles - 1) / tiles;
  splits = std::max(1, std::min(splits, nsteps / 4));
  if (splits > 1 && !a.accumulate) {
    // partial tiles are added atomically: start from zero (dW rows may be strided by ldw)
    if (hipMemset2DAsync(a.dW, (size_t)a.ldw * 4, 0, (size_t)a.K * 4, a.N, stream) != hipSuccess) return -4;
  }
  const dim3 grid(a.N / 64, a.K / 64, splits), block(256);
  const size_t lds = 6 * TILE_B;
  if (op_dtype == OP_F16) hipLaunchKernelGGL(gemm_tn_kernel<f16>, grid, block, lds, stream, a);
  else hipLaunchKernelGGL(gemm_tn_kernel<bf16>, grid, block, lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_gemm_tn_group(const GemmTnArgs* jobs, int njobs, int op_dtype, hipStream_t stream) {
  if (njobs < 1 || njobs > GEMM_TN_MAX_JOBS) return -1;
  if (njobs == 1) return launch_gemm_tn(jobs[0], op_dtype, stream);
  GemmTnGroup g;
  g.njobs = njobs;
  int tiles = 0, kmax = 0, min_steps = 1 << 30, nb = 0;
  for (int i = 0; i < njobs; ++i) {
    const GemmTnArgs& a = jobs[i];
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.N % 64 || a.K % 64) return -1;
    if (a.yv.rpi <= 0 || a.xv.rpi <= 0 || (a.yv.ld & 7) || (a.xv.ld & 7)) return -1;
    g.j[i] = a;
    g.nb_begin[i] = nb;
    nb += a.N / 64;
    tiles += (a.N / 64) * (a.K / 64);
    kmax = std::max(kmax, a.K / 64);
    min_steps = std::min(min_steps, (a.M + BMT - 1) / BMT);
  }
  for (int i = njobs; i < GEMM_TN_MAX_JOBS; ++i) { g.j[i] = jobs[0]; g.nb_begin[i] = nb; }
  g.nb_begin[GEMM_TN_MAX_JOBS] = nb;
  // the contraction split: enough workgroups for ~4 per CU, at least four steps each; one factor for all jobs
  int splits = (1024 + tiles - 1) / tiles;
  splits = std::max(1, std::min(splits, min_steps / 4));
  if (splits > 1)
    for (int i = 0; i < njobs; ++i)
      if (!jobs[i].accumulate) return -1;
  const dim3 grid(nb, kmax, splits), block(256);
  const size_t lds = 6 * TILE_B;
  if (op_dtype == OP_F16) hipLaunchKernelGGL(gemm_tn_group_kernel<f16>, grid, block, lds, stream, g);
  else hipLaunchKernelGGL(gemm_tn_group_kernel<bf16>, grid, block, lds, stream, g);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_colsum(const void* dY, long long block_stride, RowView yv, int M, int N, float* db, int accumulate, int op_dtype,
                  hipStream_t stream) {
  if (M <= 0 || N <= 0 || N % 64 || yv.rpi <= 0 || (yv.ld & 7)) return -1;
  const dim3 grid(N / 64, (M + COLSUM_ROWS - 1) / COLSUM_ROWS);
  if (grid.y > 1 && !accumulate && hipMemsetAsync(db, 0, (size_t)N * 4, stream) != hipSuccess) return -4;
  if (op_dtype == OP_F16) hipLaunchKernelGGL(colsum_kernel<f16>, grid, dim3(256), 0, stream, dY, block_stride, yv, M, db, accumulate);
  else hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, stream, dY, block_stride, yv, M, db, accumulate);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace mra
