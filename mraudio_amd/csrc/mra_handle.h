// Private to the library: the handle behind include/mra.h, its parameter registry and small host helpers
// shared by the inference path (mra_abi.hip) and the training path (mra_train.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mra.h"
#include "kernels.h"

namespace mra_host {

using namespace mra;

extern thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(MRA_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Param {
  void* ptr = nullptr;   // destination inside the arena
  int dtype = MRA_F32;   // stored dtype
  long long numel = 0;   // elements expected from the source tensor
  bool loaded = false;
  size_t goff = 0;       // byte offset of this parameter's f32 gradient in the flat gradient buffer
  int rows = 0, cols = 0;  // matrix shape ([out, in]) for weights, 0 otherwise
  float* copy32 = nullptr;  // f32 copy kept beside the operand-dtype one (the score-chain weights of the split-precision cross-attention)
};

struct LayerW {
  void *wqkv, *wo, *wcq, *wco, *wiq, *woq, *wit, *wot;
  float *bqkv, *bo, *bcq, *bco, *biq, *boq, *bit, *bot;
  float *ln1g, *ln1b, *lncg, *lncb, *lnqg, *lnqb, *lntg, *lntb;
  float* wcq32;     // f32 copy of the cross-attention query weight (split-precision cross-attention), cross layers only
  int cross_index;  // -1 when the layer has no cross-attention
  // transposed copies for the data-gradient GEMMs (training only; nullptr until mra_qformer_enable_training)
  void *wqkvT, *woT, *wcqT, *wcoT, *wiqT, *woqT, *witT, *wotT;
};

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <typename T>
  T* take(size_t count, size_t elem = sizeof(T)) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align_up(count * elem);
    return p;
  }
};

inline RowView plain(int rows, int ld) { return RowView{0, rows > 0 ? rows : 1, ld}; }
inline RowView items_view(long long item_stride, int rpi, int ld) { return RowView{item_stride, rpi, ld}; }
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int chk(int rc, const char* what) {
  if (rc == 0) return 0;
  return fail(rc == -1 ? MRA_EINVAL : MRA_EHIP, std::string(what) + " failed (rc " + std::to_string(rc) + ")");
}

}  // namespace mra_host

struct mra_qformer {
  mra_cfg cfg;
  int device = 0;
  int ncross = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::map<std::string, mra_host::Param> params;
  std::vector<mra_host::LayerW> layers;
  // embeddings / extras
  float *word = nullptr, *pos = nullptr, *embg = nullptr, *embb = nullptr, *query = nullptr;
  float *encg = nullptr, *encb = nullptr;
  void* wkv = nullptr;  // [ncross*2*H, E]
  float* bkv = nullptr;
  void* wllm = nullptr;
  float* bllm = nullptr;
  // folded cross-attention (mra_qformer_set_cross_mode): per cross layer the key weight regrouped as [heads][E][64]
  char* arena_f = nullptr;
  bool fold_stale = true;
  bool split_softmax = true;                      // scores GEMM writes exp2(s - tile max) + tile statistics (false: fp32 rows + row softmax)
  bool inreg_rescale = true;                      // split softmax: the P . enc GEMM applies the row factors to its P~ fragments (false: a rescale pass over P; cross mode 5)
  int sc_tile = 5;                                // scores: 5 = the 176 x 384 tile
  bool pv_kmajor = true;                          // P . enc reads the encoder tokens themselves (K-major weights, no enc^T copy)
  int pv_tile = 5;                                // P . enc: 5 = the 176 x 384 loader-wave tile (one workgroup per CU at E = 1408)
  int fold_tile = 2;                              // GemmProb::tile_cfg of the two batched GEMMs (2 = 128 x 128, 4 = 128 x 384)
  bool fold_stream = false;                       // folded path on the streaming kernels of fold_stream.hip (mra_qformer_set_cross_mode 4)
  int cross_mode = 0;                             // 0 automatic, 1 K/V cache, 2 folded
  // split-precision cross-attention (mra_qformer_set_cross_precision): hidden state, W_cq, Q, W_k and Q' of the score chain as operand-dtype
  // hi + lo pairs (folded form forced); wk32 = f32 copies of the key weights [ncross][H][E], arena_p = per cross layer W_cq as
  // [H][3H] (hi | hi | lo) and W_k as [heads][E][192] (hi | hi | lo), allocated when the mode is first enabled
  // layer-chain GEMMs on the ring kernel's exact-fit tiles (mra_qformer_set_option "chain_ring"): bit 0 QKV (144 x 128), bit 1 FFN-up (192 x 128),
  // bit 2 the N = hidden projections with a residual (96 x 64), bit 3 (with bit 2) their LayerNorm inside the same launch (EPI_RES_LN: measured
  // SLOWER in the step -- 6.95 vs 6.73-6.93 ms, reference item shape 2.86-2.92 vs 2.65-2.83 -- opt-in); chosen per launch only where the tile
  // divides N and the launch has >= ~1 k rows
  int chain_ring = 7;    // measured in the step (r03d, same box): mask 0 / 1 / 3 / 7 = 6.65 / 6.64 / 6.59 / 6.56 ms; reference item shape 2.55 / 2.62 / - / 2.52 ms
  int train_ring = 4;   // the same mask for the training forward / backward GEMMs (mra_qformer_set_option "train_ring"): bit 0 QKV, bit 2 every N = hidden
                        // GEMM whose epilogue the ring kernel has (projections with a residual, the data gradients).  Measured at B = 1 x T = 20 (r03x,
                        // one session): 14.6-15.2 ms per step with 0, 14.0-14.3 with 4, 14.2 with 5
  int cross_precise = 0;
  float* wk32 = nullptr;
  char* arena_p = nullptr;
  bool precise_stale = true;
  hipEvent_t kv_done = nullptr;                   // optional scheduling hook (mra_qformer_set_kv_done_event)
  hipEvent_t kv_ev0 = nullptr, kv_ev1 = nullptr;  // optional instrumentation (mra_qformer_set_kv_events)
  // training
  char* arena_t = nullptr;      // transposed weight copies
  bool transposes_stale = true;
  mra::TrJob* tr_jobs = nullptr;   // device table of the batched transpose (built on first use)
  int n_tr_jobs = 0, n_tr_tiles = 0;
  hipStream_t wg_stream = nullptr;                // side stream of the weight-gradient GEMMs (mra_qformer_backward)
  hipEvent_t wg_ev[32] = {};                      // ring of fork / done events between the caller's stream and wg_stream
  // fused optimizer pass (mra_qformer_adam_step): matrices with a transposed training copy as 64 x 64 tile jobs, every other bert.* parameter
  // (and the query tokens) as linear segments, the f32 copies of the score-chain weights as a third table; built by mra_qformer_enable_training
  mra::AdamMatJob* adam_jobs = nullptr;
  int n_adam_jobs = 0, n_adam_tiles = 0;
  mra::FlatSeg* adam_segs = nullptr;
  int n_adam_segs = 0;
  mra::FlatSeg* c32_segs = nullptr;
  int n_c32_segs = 0;
  size_t grad_bytes = 0;
  mra::FlatSeg* flat_segs = nullptr;  // device table behind mra_qformer_load_flat (bert.* parameters)
  int n_flat_segs = 0;
  int op() const { return cfg.op_dtype == MRA_BF16 ? mra::OP_BF16 : mra::OP_F16; }
};

namespace mra_host {
// folded cross-attention pays once the encoder sequence is long (fewer flops at any Kv, but five launches per layer)
inline bool use_fold(const mra_qformer* h, int kv) { return h->cross_precise || h->cross_mode == 2 || (h->cross_mode == 0 && kv >= 2048); }
// padded score-row length: whole 128- and 176-row tiles of the scores GEMM, and a multiple of 128 (K of P . enc)
// P . enc on the 176 x 384 tile with K-major weights: no transposed copy of the encoder tokens is needed
inline bool fold_kmajor(const mra_qformer* h) {
  return h->pv_kmajor && h->pv_tile == 5 && h->cfg.heads * h->cfg.n_query == 384 && h->cfg.enc_width % 176 == 0;
}
// split softmax without the rescale pass: needs the 176-column score tiles and the K-major 176 x 384 P . enc tile
inline bool fold_inreg_rescale(const mra_qformer* h) { return h->inreg_rescale && h->split_softmax && h->sc_tile == 5 && fold_kmajor(h); }
inline int fold_kvp(int kv) { return (std::max((kv + 127) / 128 * 128, (kv + 175) / 176 * 176) + 127) / 128 * 128; }
// the streaming kernels (fold_stream.hip): f16 operands, 384 (head, query) rows, E a multiple of 176
inline bool fold_streams(const mra_qformer* h, int kv) {
  return !h->cross_precise && h->fold_stream && mra::fold_stream_supported(h->cfg.heads * h->cfg.n_query, h->cfg.enc_width, kv, fold_kvp(kv), h->op());
}
// split-precision cross-attention: bytes of one cross layer's prepared weights, W_cq [H][3H] then W_k [heads][E][192] (operand dtype)
inline size_t precise_wk_off(const mra_qformer* h) { return align_up((size_t)h->cfg.hidden * 3 * h->cfg.hidden * 2); }
inline size_t precise_layer_bytes(const mra_qformer* h) {
  return precise_wk_off(h) + align_up((size_t)h->cfg.heads * h->cfg.enc_width * 192 * 2);
}
// K/V of every cross layer in ONE GEMM: [items*kv, E] x [ncross*2*H, E]^T, scattered head-major.
int kv_project(const mra_qformer* h, const void* enc, int N, int kv, void* kv_cache, hipStream_t stream);
}  // namespace mra_host
