// Host-visible launchers of the gfx950 kernels.  Everything takes raw device pointers and a
// stream; nothing here allocates, synchronises or touches torch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mra {

enum OpDtype { OP_F16 = 0, OP_BF16 = 1 };

// GEMM epilogues (C index is [activation row m][weight row n])
enum GemmEpi {
  EPI_OP = 0,       // C(op dtype) = acc + bias
  EPI_GELU_OP = 1,  // C(op dtype) = gelu_erf(acc + bias)
  EPI_RES_F32 = 2,  // C(f32)      = acc + bias + R
  EPI_F32 = 3,      // C(f32)      = acc + bias
  EPI_KV = 4,       // C(op dtype) head-major K/V cache, see GemmProb::kv_*
  EPI_GELU_BOTH = 6, // training forward: aux(op dtype) = acc + bias (the pre-activation the backward needs), C(op dtype) = gelu_erf of it
  EPI_GELU_BWD = 7,  // training backward: C(op dtype) = acc * gelu'(aux): the GELU gradient inside the data-gradient GEMM
  EPI_RES_OP = 8,   // C(op dtype) = op(acc + bias) + Rop: residual stream kept in the operand dtype (the ViT under precision="fp16");
                    // Rop is GemmProb::aux (op dtype, addressed like C; may alias C: updated in place)
  EPI_RES_LN = 9,   // EPI_RES_F32 followed by the LayerNorm of the finished rows INSIDE the launch (ring tiles only): the column tile of a row block
                    // that finishes last (agent-scope counter GemmProb::ln_counter, one per row tile, zero before the launch, reset by the kernel)
                    // normalises the block's rows and writes ln_y32 / ln_y16 -- the reference's `LayerNorm(dense(x) + residual)` (HF:519-530,
                    // 573-587) in one launch.  C still receives the pre-LayerNorm rows (sc1 stores: other XCDs read them).
  // LayerNorm folded into the GEMMs on either side of it (the ViT blocks, fp32 residual stream): y = LN(x) W^T + b with LN's gain inside
  // W' = W diag(g) is  rstd[m] * (x W'^T - mu[m] * colsum(W')[n]) + (b + W beta)[n]  -- a GEMM over the op-dtype copy of the RAW rows x plus a
  // rank-one correction in the epilogue.  The producer of x (a residual GEMM) emits that copy and the row statistics as it stores x:
  EPI_RES_F32_STAT = 10,  // EPI_RES_F32, plus: ln_y16 (op dtype, row view ln_y16v) = the finished rows, and per (row, 128-column group) the
                          // group mean and the group's sum of squared deviations, ln_y32[(m * (N / 128) + n / 128) * 2 + {0, 1}]
                          // (launch_ln_group_stats folds the groups into (mean, rstd) per row).  Eight-phase tiles only (a wave owns 128 columns).
  EPI_RES_OP_STAT = 13,   // EPI_RES_OP (residual stream in the operand dtype), plus the statistics of the ROUNDED rows per (row, 64-column block):
                          // ln_y32[(m * (N / 64) + n / 64) * 2 + {0, 1}]; the stream itself is the consumer's operand, no copy.  Eight-phase tiles only.
  EPI_LNF_OP = 11,        // C(op dtype) = rstd[m] * (acc - mu[m] * ln_gain[n]) + bias[n]; (mu, rstd)[m] = ln_y32[2 m], ln_y32[2 m + 1]; ln_gain = colsum(W')
  EPI_LNF_GELU_OP = 12,   // gelu_erf of the same (256 x 256 eight-phase tile only, both)
  EPI_SOFTPART = 5, // C(op dtype) = exp2(alpha * acc - max over the tile's columns of the row); the row's tile maximum and tile
                    // sum go to stat_m / stat_l [row][ntiles] (176 x 384 loader-wave tile only: a wave holds whole tile rows)
};

// Logical activation row m of a "row view" lives at base + (m / rpi) * item_stride + (m % rpi) * ld
// (elements).  This lets the 32 query rows or the L text rows of every [S, H] item be addressed in
// place, without gathers.
struct RowView {
  long long item_stride;
  int rpi;  // rows per item
  int ld;   // row stride
};

// One problem of a (possibly grouped) launch: C[m][n] = sum_k A[m][k] * W[n][k] (+ epilogue).
struct GemmProb {
  const void* A;      // activations, op dtype, row view `a`
  const void* W;      // weights [N][K] row-major (the reference's nn.Linear layout), op dtype
  const float* bias;  // [N] or nullptr
  void* C;            // row view `c`
  const float* R;     // residual (EPI_RES_F32), row view `r`
  void* aux;          // EPI_GELU_BOTH (written) / EPI_GELU_BWD (read): op-dtype tensor addressed like C (row view `c`)
  RowView a, c, r;
  int M, N, K;
  // EPI_KV: weight row n = (cl * 2 + kv) * hidden + head * 64 + d ; activation row m = item * kv_tokens + tok
  // destination element ((((cl * 2 + kv) * kv_items + item) * kv_heads + head) * kv_tokens + tok) * 64 + d
  int kv_tokens, kv_items, kv_heads;
  // Batched launch (batch > 1): problem b uses A + b * a_bs, W + b * w_bs (elements), bias + b * bias_bs and
  // C + b * c_bs_bytes; M, N, K and the row views are shared.  Tiles never straddle batch entries.
  int batch;
  long long a_bs, w_bs, c_bs_bytes;
  int bias_bs;
  // n_ragged: N need not be a multiple of the tile; weight rows past N - 1 are read from row N - 1 and their
  // output columns are written anyway, so C rows must hold ceil(N / tile) * tile columns and bias must be null.
  int n_ragged;
  // n_mask (EPI_RES_F32 / EPI_F32): N need not be a multiple of the tile either, but output columns past N - 1 are simply not
  // stored (bias and residual are not read there): C keeps its plain [M][N] rows.  N % 4 == 0.
  int n_mask;
  // w_ld > 0 (176 x 384 loader-wave tile, EPI_OP only): W is given K-major, W[k][n] at W + k * w_ld + n (a plain [K][N] row-major
  // matrix, e.g. the encoder tokens themselves for P . enc); rows k >= k_rows are read from row k_rows - 1 (the A operand
  // is zero there).  The kernel stages [64 k][176 n] tiles and takes its fragments with ds_read_b64_tr_b16.
  int w_ld, k_rows;
  // EPI_RES_LN: LayerNorm over the N columns of every finished row (N = 256 k <= 1024): y = (x - mean) * rstd * ln_gain + ln_bias, biased variance,
  // fp32; outputs ln_y32 (fp32, row view ln_y32v; may be null) and ln_y16 (operand dtype, row view ln_y16v; may be null)
  const float* ln_gain;
  const float* ln_bias;
  float ln_eps;
  float* ln_y32;
  void* ln_y16;
  RowView ln_y32v, ln_y16v;
  unsigned* ln_counter;   // [ceil(M / tile rows)] zero before the launch; the kernel leaves it zero
  // w_kwrap > 0 (loader-wave kernels, row-major W): W holds only w_kwrap K steps of 64 (K = 2 * 64 * w_kwrap) and is walked twice -- C = [A_hi | A_lo] . [W | W]^T,
  // the scores product of the split-precision cross-attention (Q' kept as an f16 hi + lo pair against the same encoder slab)
  int w_kwrap;
  // EPI_SOFTPART: scale of the scores in log2 units and the per-(row, column tile) statistics, row m of batch entry b at
  // stat_*[(b * M + m) * ntiles + tile]
  float alpha;
  float* stat_m;
  float* stat_l;
  // pscale != nullptr (176 x 384 loader-wave tile with K-major W only): the A operand is the unnormalised P~ of the split softmax; A[m][k]
  // is multiplied by pscale[((batch entry * ps_ntiles) + min(k / 176, ps_ntiles - 1)) * 512 + m] on its way into the MFMA (f16: packed
  // multiplies with the factor rounded to f16; bf16: in fp32, rounded once -- what the separate rescale pass did in HBM).  A must be
  // ZERO from column ps_ntiles * 176 on (launch_fold_rowfactor does that).  M <= 384 (one row tile: the kernel indexes the
  // 512-row factor slice by the row inside the tile).
  const float* pscale;
  int ps_ntiles;
  int tile_cfg;    // 0 = automatic, 1 / 2 / 3 = force the 64 / 128 / 256 tile, 4 = the 128 (weight rows) x 384 (activation
                   // rows) loader-wave tile (EPI_OP / EPI_F32 only), 5 = 176 x 384 with the compute waves in one column (EPI_OP only,
                   // N % 176 == 0), 6 = 64 (weight rows) x 128 (activation rows) with 128-deep K steps (K % 128 == 0: the long-K down-projections of
                   // the layer chain at ~1 k rows, where a tile's bytes per flop, not its count, sets the time), 7 = 128 (weight rows) x 512
                   // (activation rows) on the eight-phase kernel (N = 128 exactly: the half-width last column tile of N = 256 k + 128, launched on its
                   // own with W / bias / C / R offset to those columns; EPI_RES_F32 / EPI_F32 / EPI_RES_OP, K % 128 == 0), 8 = N = 256 k + 128 in ONE launch:
                   // full 256 x 256 eight-phase tiles plus one 128 x 512 tail tile per pair of row tiles (same epilogues, one problem); the first problem decides;
                   // 9 / 10 / 11 = the ring kernel's exact-fit tiles of the layer chain: 144 (weight rows) x 128 (activation rows), 192 x 128, 96 x 64 -- one
                   // workgroup per CU at 2048 x 2304, 2 x 1024 x 3072 and 2048 x 768 (plain problems, N a multiple of the tile, EPI_OP / GELU_OP / RES_F32 / F32)
  int order;       // tile walk: 0 = panels of 8 row tiles, rows fastest; gn > 0 = panels of gn column tiles walked down the rows, columns
                   // fastest (measured better for the ViT's N = 1408 GEMMs: all 6 column tiles of a row tile run together)
  int persist;     // eight-phase 256 x 256 kernel, staged epilogues: 1 = ONE workgroup per CU walks over the tiles (blockIdx, + gridDim, ...) instead of one
                   // workgroup per tile.  A stamped timeline (gemm_bench p8stamp, fc1 shape) shows ~3.8 us between the end of a workgroup and the start
                   // of the next on its CU plus ~1.8 us of kernel entry / argument fetch / address set-up per workgroup, of a ~50 us tile
  int tile_begin;  // filled by the launcher
  int mtiles, ntiles;
  int batch_row0;  // filled per workgroup: first row of its batch entry in the EPI_SOFTPART statistics
};

constexpr int GEMM_MAX_GROUPS = 4;   // problems per launch (round 3: the pair forward groups the query / text problems of two Q-Formers)
struct GemmArgs {
  GemmProb p[GEMM_MAX_GROUPS];
  int ngroups;
  int total_tiles;
  unsigned long long* dbg;  // diagnostic builds only (gemm_set_debug_buffer): per-wave cycle sums
  int order;                // 0: 8-row-tile panels, rows fastest; gn > 0: panels of gn column tiles, columns fastest (GemmProb::order of the first problem)
};

// Returns 0 on success, <0 on bad shapes.  All problems of one launch share dtype / epilogue.
int launch_gemm(const GemmProb* probs, int ngroups, int epi, int op_dtype, hipStream_t stream);
// which tile config launch_gemm would pick (for tests / DESIGN.md): 0 = 64x64, 1 = 128x128, 2 = 256x256
int gemm_pick_config(const GemmProb* probs, int ngroups);
// Main loops ("families") launch_gemm dispatches to, and a read-only launch counter per (family, epilogue) since the library was
// loaded (mra_debug_gemm_launches in include/mra.h): parity tests use it to state which kernel produced the numbers they checked.
enum GemmFamily {
  GF_V1_64 = 0, GF_V1_128 = 1, GF_V1_256 = 2,   // gemm_kernel two-buffer loop (256: experiment builds only)
  GF_WS_256 = 3,                                // gemm_ws_kernel 256 x 256 (odd number of K steps)
  GF_P8_256 = 4,                                // gemm_p8_kernel eight-phase 256 x 256
  GF_WS_128x384 = 5, GF_WS_176x384 = 6,         // loader-wave tiles of the folded cross-attention
  GF_K128_64x128 = 7,                           // gemm_k128_kernel, 128-deep steps
  GF_P8_TAIL = 8, GF_P8_MIXED = 9,              // eight-phase 128 x 512 tail tile; full tiles + tail tile in one launch
  GF_K128_64x64 = 10,
  GF_RING_144x128 = 11, GF_RING_192x128 = 12, GF_RING_96x64 = 13,   // gemm_ring_kernel: exact-fit tiles of the layer chain, K tiles through an LDS ring
  GEMM_FAMILIES = 14
};
long long gemm_launch_count(int family, int epi);
#ifdef MRA_GEMM_EXPERIMENTS
// Process-global A/B switches: compiled ONLY into tests/native/libmra_hip_exp.so (the experiment library of tests/native/gemm_bench and
// kernel_check_exp); the shipped libmra_hip.so has no mutable global state besides the launch counters above.
void gemm_force_config(int cfg);  // -1 = automatic (default)
void gemm_set_debug_buffer(unsigned long long* dev_buf);  // variant 4 (stamped v1) writes 4 u64 per wave
void gemm_set_eight_phase(int on);         // 256 x 256 launches with an even number of K steps on gemm_p8_kernel (A/B switch)
void gemm_set_tile_order(int order);      // overrides GemmProb::order for every later launch when != 0 (A/B runs)
void gemm_force_variant(int v);   // 5 = default (warp-specialised 256x256, two-buffer small tiles); 0 ring, 1 two-buffer,
                                  // 2 +L2 prefetch, 3 +spread DMA issue, 4 stamped diagnostic -- kept for A/B runs
#endif

// ---- backward-pass GEMMs ------------------------------------------------------------------------
// dW[n][k] (+)= sum_m dY[m][n] * X[m][k].  dY and X are given as 64-column blocks: block b of dY starts at
// dY + b * y_block_stride and its row m at + row-view(yv, m) (row-major [M, N]: block stride 64, ld N;
// head-major K/V gradients: block = (layer, k|v, head), stride kv * 64, item stride heads * kv * 64, ld 64).
struct GemmTnArgs {
  const void* dY;
  const void* X;
  float* dW;   // [N][ldw] f32
  RowView yv, xv;
  long long y_block_stride, x_block_stride;
  int M, N, K;
  int ldw;
  int accumulate;  // 1: dW += ...
  float* db;       // optional bias gradient db[n] += sum_m dY[m][n], always accumulated (the column-block-0 workgroups
                   // multiply their dY fragments by a ones operand: two more MFMAs per step instead of a second launch)
};
int launch_gemm_tn(const GemmTnArgs& a, int op_dtype, hipStream_t stream);
// Up to GEMM_TN_MAX_JOBS weight gradients in ONE launch (the weight gradients that become computable at the same point of the backward:
// query / key / value / output of a self-attention, the two matrices of a feed-forward, ...).  A weight gradient of the layer chain is ~150
// output tiles of a few hundred contraction rows: a launch of its own is a chip full of short, latency-bound workgroups (21 us on average, 240
// of them per training step); launched together the jobs' workgroups share the CUs.  All jobs must be accumulating (dW += ...) when the
// contraction is split (it is split by the same factor for every job).
constexpr int GEMM_TN_MAX_JOBS = 4;
struct GemmTnGroup {
  GemmTnArgs j[GEMM_TN_MAX_JOBS];
  int njobs;
  int nb_begin[GEMM_TN_MAX_JOBS + 1];   // filled by the launcher: first 64-row block of dW (grid.x index) of every job
};
int launch_gemm_tn_group(const GemmTnArgs* jobs, int njobs, int op_dtype, hipStream_t stream);
// db[n] (+)= sum_m dY[m][n]
int launch_colsum(const void* dY, long long block_stride, RowView yv, int M, int N, float* db, int accumulate, int op_dtype,
                  hipStream_t stream);

// ---- backward-pass row / elementwise / attention kernels (backward.hip) ----------------------------
struct LnBwdArgs {
  const float* dy; RowView dyv;   // upstream gradient of the LayerNorm output
  const float* x; RowView xv;     // saved pre-LN rows (statistics are recomputed)
  const float* gamma;
  float eps;
  int rows;
  float* dx; RowView dxv;         // gradient w.r.t. the pre-LN rows (+ `add` when given)
  const float* add; RowView addv; // optional second gradient stream added into dx (the residual branch)
  void* dx16; RowView dx16v;      // optional operand-dtype copy of dx
  float* dgamma; float* dbeta;    // optional, accumulated with float atomics
};
int launch_ln_bwd(const LnBwdArgs& a, int H, int op_dtype, hipStream_t stream);
// two independent LayerNorm backward jobs in one launch (b may be null or empty): the query-row and the text-row LayerNorm of a Q-Former layer
int launch_ln_bwd2(const LnBwdArgs& a, const LnBwdArgs* b, int H, int op_dtype, hipStream_t stream);
// out = gelu_erf(u) (backward = 0) or out = df * gelu'(u) (backward = 1); operand dtype, n % 8 == 0
int launch_gelu(const void* u, const void* df, void* out, long long n, int backward, int op_dtype, hipStream_t stream);
struct AttnBwdArgs {
  const void *Q, *K, *V, *O, *dO;   // operand dtype; O / dO share the [item][q row][head*64 + d] layout
  void *dQ, *dK, *dV;               // operand dtype; same layouts as Q / K / V (own strides)
  long long q_item_stride, o_item_stride, dq_item_stride;
  int q_ld, o_ld, dq_ld;
  long long k_item_stride, k_head_stride, v_item_stride, v_head_stride;
  int k_ld, v_ld;
  long long dk_item_stride, dk_head_stride, dv_item_stride, dv_head_stride;
  int dk_ld, dv_ld;
  const long long* mask; int mask_ld;
  const float* lse;                 // [item][head][q_rows] from the forward
  int items, heads, q_rows, kv_len;
  float scale;
};
int launch_attn_bwd(const AttnBwdArgs& a, int op_dtype, hipStream_t stream);
#ifdef MRA_GEMM_EXPERIMENTS
void attn_bwd_force_valu(int on);   // A/B switch (experiment library only): the first (fp32 VALU) kernel instead of the MFMA one
#endif
int launch_embed_bwd(const float* demb, const long long* ids, int items, int L, int Q, int H, int vocab, float* dquery, float* dpos,
                     float* dword, hipStream_t stream);
int launch_transpose16(const void* src, void* dst, int R, int C, int op_dtype, hipStream_t stream);
// many transposes in one launch: job j owns the 32x32 tiles [tile_begin, next job's tile_begin) of its matrix
struct TrJob { const void* src; void* dst; int R, C, tile_begin, tiles_x; };
int launch_transpose16_batch(const TrJob* jobs_dev, int njobs, int total_tiles, int op_dtype, hipStream_t stream);
int launch_add_f32(const float* x, float* y, long long n, hipStream_t stream);  // y += x
struct FlatSeg;
// Fused optimizer pass of the training step (backward.hip): torch.optim.Adam's update on the flat fp32 master / gradient / moment buffers AND
// the refresh of the device weights it makes stale, in one read of each: per element  g += wd p;  m = b1 m + (1 - b1) g;
// v = b2 v + (1 - b2) g^2;  p -= step_size m / (sqrt(v) inv_sqrt_bc2 + eps);  then p goes out in the stored dtype (and, for the matrices
// with a transposed training copy, transposed as well through a 64 x 64 LDS tile); zero_grad clears g on the way.
struct AdamScalars { float beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2; int zero_grad; };
struct AdamMatJob { unsigned long long off; void* dst; void* dstT; int R, C, ldT, tile_begin, tiles_x; };   // master offset (elements); dstT[c * ldT + r]
int launch_adam_mats(float* master, float* grad, float* m, float* v, const AdamMatJob* jobs_dev, int njobs, int total_tiles, AdamScalars sc, int op_dtype,
                     hipStream_t stream);
int launch_adam_flat(float* master, float* grad, float* m, float* v, const FlatSeg* segs, int nseg, AdamScalars sc, hipStream_t stream);

// ---- folded cross-attention helpers (fold.hip) ------------------------------------------------------
// P[row][0..kv) = softmax(scale * S[row][0..kv)) in the operand dtype, P[row][kv..kvp) = 0
int launch_softmax_rows(const float* S, long long ld_s, void* P, long long ld_p, int rows, int kv, int kvp, float scale, int op_dtype,
                        hipStream_t stream);
// second half of the split softmax (first half: EPI_SOFTPART of the scores GEMM): row statistics over the column tiles,
// then P[row][k] *= exp2(m_tile - m_row) / sum in place; columns [kv_cols, kvp) are zeroed.  tile_cols = columns per tile.
// split softmax, second half without a pass over P: factors[(row / R * ntiles + t) * 512 + row % R] = exp2(m_tile - m_row) / L_row from the
// tile statistics stat_*[row * ntiles + t] (rows = items * R, R <= 512); read by the P . enc GEMM through GemmProb::pscale.  Also zeroes
// the 16-bit P~ rows (row stride ld_p elements) from column ntiles * tile_cols to kvp, which the scores GEMM leaves unwritten.
int launch_fold_rowfactor(const float* stat_m, const float* stat_l, float* factors, int rows, int R, int ntiles, void* P, long long ld_p, int tile_cols,
                          int kvp, hipStream_t stream);
int launch_softmax_rescale(void* P, long long ld_p, const float* stat_m, const float* stat_l, int rows, int ntiles, int tile_cols, int kvp,
                           int op_dtype, hipStream_t stream);
// dst[b][c][r] = src[b][r][c] (r < R), 0 for R <= r < ld_d; src [batch][R][C], dst [batch][C][ld_d]
int launch_transpose_pad(const void* src, void* dst, int R, int C, int ld_d, long long src_bs, long long dst_bs, int batch, int op_dtype,
                         hipStream_t stream);

// ---- folded cross-attention, streaming form (fold_stream.hip) ---------------------------------------
// scores + split-softmax statistics + P . enc of one cross layer for `items` items of 384 (head, query) rows:
//   phase bit 0: P~ = exp2(alpha * Q' enc^T - ceil(tile max)) in f16 + per-(row, 176-column tile) statistics, then the row
//                statistics (tile factors 2^(m_tile - m_row) as f16, 1 / L);   phase bit 1: U = (1 / L) sum g P~ enc.
struct FoldStreamArgs {
  const void* qp;    // Q' [items][384][E] f16, row-major (the per-head Q' GEMM's output)
  void* qpb;         // workspace of the same size: Q' re-packed into the row operand's blocked layout
  const void* enc;   // encoder tokens [items][kv][E] f16
  void* p;           // P~ [items][384 x kvp] f16 (workspace; blocked layout [row / 16][k / 8][16][8])
  void* u;           // U  [items][384][E] f16
  float* stat_m;     // [items * 384][fold_stream_stat_ld(kvp)]
  float* stat_l;
  void* gexp;        // f16 [items * 384][stat_ld]
  float* ginv;       // [items * 384]
  int items, kv, kvp, E;
  float alpha;
  int phase;         // 3 = both
};
bool fold_stream_supported(int rows, int E, int kv, int kvp, int op_dtype);
int fold_stream_stat_ld(int kvp);
int launch_fold_stream(const FoldStreamArgs& a, hipStream_t stream);

// ---- attention ------------------------------------------------------------------------------
struct AttnArgs {
  const void* Q;  // [item][q row][head*64 + d], op dtype
  const void* K;  // token rows of 64 elements
  const void* V;
  void* O;        // [item][q row][head*64 + d], op dtype
  long long q_item_stride, o_item_stride;  // elements
  int q_ld, o_ld;
  long long k_item_stride, k_head_stride;  // elements; token row stride = k_ld
  long long v_item_stride, v_head_stride;
  int k_ld, v_ld;
  const long long* mask;  // [item][mask_ld] int64 (1 = attend) or nullptr; additive (1-m) * -10000
  int mask_ld;
  int items, heads, q_rows, kv_len;
  float scale;   // 1/sqrt(64)
  // grid-level KV split (cross attention with long KV): partials in `part`, combined by a 2nd kernel
  int nsplit;
  float* part;   // [items*heads*qblocks][nsplit][32*64 + 64] f32
  float* lse;    // optional [item][head][q_rows] f32: log2-sum-exp2 of the scaled, masked scores (for the backward)
};
size_t attn_partial_bytes(int items, int heads, int q_rows, int nsplit);
int attn_pick_split(int items, int heads, int q_rows, int kv_len);
int launch_attention(const AttnArgs& a, int op_dtype, hipStream_t stream);

// ---- normalisation / embeddings / conversions -----------------------------------------------
// y = LN(x) over H (multiple of 256, <= 1024) with gain/bias; writes f32 and/or op-dtype copies.
int launch_ln_rows(const float* x, RowView xv, int rows, int H, const float* gain, const float* bias, float eps,
                   float* y32, RowView y32v, void* y16, RowView y16v, int op_dtype, hipStream_t stream);
// as launch_ln_rows, but rows with (row % period) >= split use (gain2, bias2): the query and the text LayerNorm
// of a Q-Former layer in one launch over all rows of the [items, 32 + L, H] stream
int launch_ln_rows2(const float* x, RowView xv, int rows, int H, const float* gain, const float* bias, const float* gain2,
                    const float* bias2, int period, int split, float eps, float* y32, RowView y32v, void* y16, RowView y16v,
                    int op_dtype, hipStream_t stream);
// two Q-Formers in one launch (pair forward): rows [0, lane_rows) as launch_ln_rows2 with sets (1, 2), rows from lane_rows on with sets (3, 4)
// (gain4 == nullptr: set 3 for all of them)
int launch_ln_rows4(const float* x, RowView xv, int rows, int H, const float* gain, const float* bias, const float* gain2, const float* bias2,
                    int lane_rows, const float* gain3, const float* bias3, const float* gain4, const float* bias4, int period, int split,
                    float eps, float* y32, RowView y32v, void* y16, RowView y16v, int op_dtype, hipStream_t stream);
// Modality LayerNorm (A2) fused with the item gather (A3): out item i <- LN(in item index[i]).
// x dtype: 0 = f32, 1 = f16, 2 = bf16.
int launch_modality_ln(const void* x, int x_dtype, const long long* item_index, int items, int tokens, int E,
                       const float* gain, const float* bias, float eps, void* out, int op_dtype, hipStream_t stream);
// Embeddings (A4a): h[n, s] = LN(s < Q ? query[s] : word[ids[n, s-Q]] + pos[s-Q])
// query: [1 or items][Q][H]; query_item_stride = 0 broadcasts one set of query tokens to every item
int launch_embed_ln(const long long* ids, int items, int L, int Q, int H, int vocab, const float* query,
                    long long query_item_stride, const float* word,
                    const float* pos, const float* gain, const float* bias, float eps, float* h32, void* h16,
                    float* pre32 /* optional: rows before the LayerNorm (training) */, int op_dtype, hipStream_t stream);
// dst(op dtype)[rows][cols] at row offset <- src (0 = f32, 1 = f16, 2 = bf16)
// One launch that refreshes many parameters from a flat f32 master buffer: segment s copies / converts
// n (<= FLAT_SEG) elements from master + src_off to dst in the stored dtype (0 f32, 1 f16, 2 bf16).
struct FlatSeg { unsigned long long src_off; void* dst; int n; int dtype; };
constexpr int FLAT_SEG = 16384;
int launch_convert_flat(const float* master, const FlatSeg* segs, int nseg, hipStream_t stream);
int launch_convert(const void* src, int src_dtype, void* dst, int dst_dtype /*0 f32,1 f16,2 bf16*/, long long n,
                   hipStream_t stream);
int launch_copy_rows_f32(const float* src, RowView sv, float* dst, RowView dv, int rows, int H, hipStream_t stream);
// split-precision cross-attention: fp32 -> operand-dtype (hi, lo) pairs laid out for K-concatenated GEMMs (norm_embed.hip)
//   rows:   dst[row][c / chunk][part][c % chunk], parts = 3: (hi, lo, hi), 2: (hi, lo); dense dst rows of C * parts elements
//   weight: [rows][C] -> [rows][3 C] = (hi | hi | lo);  key weight: [heads * 64][E] -> [heads][E][192] = (hi | hi | lo) over the head dims
int launch_split_rows(const float* src, RowView sv, int rows, int C, int chunk, int parts, void* dst, int op_dtype, hipStream_t stream);
int launch_split_weight(const float* W, int rows, int C, void* dst, int op_dtype, hipStream_t stream);
int launch_split_key_weight(const float* W, int heads, int E, void* dst, int op_dtype, hipStream_t stream);

// ---- scorer -------------------------------------------------------------------------------------
int launch_cosine_score(const float* z, const float* t, int t_rows, int items, int Q, int H, float eps, float* sim,
                        float* logit, hipStream_t stream);
int launch_fuse_logits(const float* const* logits, const float* weights, int nmod, int n, float* out,
                       hipStream_t stream);
int launch_span(const float* logits, int videos, int clips, float alpha, int* spans, hipStream_t stream);

}  // namespace mra
