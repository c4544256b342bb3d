// Training path of the Q-Former: forward that keeps an activation tape, and the full backward.
//
// BASELINE config 5 ("finetune.py step: Q-Former fwd+bwd with fused LN/GELU grads").  The reference
// itself never trains its Q-Formers (frozen, models/xinstructblip.py:196-204; the trainer only updates
// LoRA adapters of the LLM, utils/trainer.py:124-140), so the definition of correct is torch.autograd
// over the CPU oracle.  Layer structure as in the inference path (mra_abi.hip / HF:591-709).
//
// Gradients: one flat f32 buffer, one block per parameter (mra_qformer_grad_offset); every call ADDS
// into it (the caller zeroes it when it wants fresh gradients), which is what gradient accumulation
// (utils/trainer.py:31,137) needs.  Data-gradient GEMMs reuse the forward GEMM kernels on transposed
// weight copies (built by mra_qformer_enable_training, refreshed after every weight upload); weight
// gradients use gemm_tn.hip; LayerNorm / GELU / attention / embedding gradients backward.hip.
#include <cmath>

#include "mra_handle.h"

using namespace mra;
using namespace mra_host;

namespace {

struct LayerBuf {
  float* hin32; char* hin16;
  char* qkv16; float* lse_s; char* ctx16;
  float* pre1; float* h1_32; char* h1_16;
  char* qc16; float* lse_c; char* cctx16; float* pre2; float* hc32; char* hc16;
  char* u16; char* f16;
  float* pre3;
};

struct TrainBufs {
  std::vector<LayerBuf> layer;
  float* emb_pre; float* out32; char* out16;
  char* kv16; float* part; int nsplit;
  // backward scratch
  float *dhA, *dhB, *dpre32, *dhc32, *dpre2_32;
  char *dpre16, *dpre16b, *dff16, *dctx16, *dqkv16, *dpre2_16, *dcctx16, *dqc16, *dkv16;
  size_t bytes;
};

TrainBufs layout_train(const mra_qformer* h, char* base, int N, int L, int Kv) {
  const mra_cfg& c = h->cfg;
  const size_t H = c.hidden, I = c.inter, S = c.n_query + L, Q = c.n_query, heads = c.heads;
  Carver cv(base);
  TrainBufs t;
  t.layer.resize(c.layers);
  for (int i = 0; i < c.layers; ++i) {
    LayerBuf& b = t.layer[i];
    b.hin32 = cv.take<float>(N * S * H);
    b.hin16 = cv.take<char>(N * S * H, 2);
    b.qkv16 = cv.take<char>(N * S * 3 * H, 2);
    b.lse_s = cv.take<float>(N * heads * S);
    b.ctx16 = cv.take<char>(N * S * H, 2);
    b.pre1 = cv.take<float>(N * S * H);
    b.h1_32 = cv.take<float>(N * S * H);
    b.h1_16 = cv.take<char>(N * S * H, 2);
    if (h->layers[i].cross_index >= 0) {
      b.qc16 = cv.take<char>(N * Q * H, 2);
      b.lse_c = cv.take<float>(N * heads * Q);
      b.cctx16 = cv.take<char>(N * Q * H, 2);
      b.pre2 = cv.take<float>(N * Q * H);
      b.hc32 = cv.take<float>(N * Q * H);
      b.hc16 = cv.take<char>(N * Q * H, 2);
    } else {
      b.qc16 = b.cctx16 = b.hc16 = nullptr;
      b.lse_c = b.pre2 = b.hc32 = nullptr;
    }
    b.u16 = cv.take<char>(N * S * I, 2);
    b.f16 = cv.take<char>(N * S * I, 2);
    b.pre3 = cv.take<float>(N * S * H);
  }
  t.emb_pre = cv.take<float>(N * S * H);
  t.out32 = cv.take<float>(N * S * H);
  t.out16 = cv.take<char>(N * S * H, 2);
  t.kv16 = cv.take<char>((size_t)h->ncross * 2 * N * Kv * H, 2);
  t.nsplit = attn_pick_split(N, c.heads, (int)Q, Kv);
  t.part = cv.take<float>(attn_partial_bytes(N, c.heads, (int)Q, t.nsplit) / 4 + 64);
  t.dhA = cv.take<float>(N * S * H);
  t.dhB = cv.take<float>(N * S * H);
  t.dpre32 = cv.take<float>(N * S * H);
  t.dhc32 = cv.take<float>(N * Q * H);
  t.dpre2_32 = cv.take<float>(N * Q * H);
  t.dpre16 = cv.take<char>(N * S * H, 2);
  t.dpre16b = cv.take<char>(N * S * H, 2);
  t.dff16 = cv.take<char>(N * S * I, 2);
  t.dctx16 = cv.take<char>(N * S * H, 2);
  t.dqkv16 = cv.take<char>(N * S * 3 * H, 2);
  t.dpre2_16 = cv.take<char>(N * Q * H, 2);
  t.dcctx16 = cv.take<char>(N * Q * H, 2);
  t.dqc16 = cv.take<char>(N * Q * H, 2);
  t.dkv16 = cv.take<char>((size_t)h->ncross * 2 * N * Kv * H, 2);
  t.bytes = cv.off;
  return t;
}

size_t layout_transposes(mra_qformer* h, char* base) {
  const mra_cfg& c = h->cfg;
  const size_t H = c.hidden, I = c.inter;
  Carver cv(base);
  for (int i = 0; i < c.layers; ++i) {
    LayerW& L = h->layers[i];
    L.wqkvT = cv.take<char>(3 * H * H, 2);
    L.woT = cv.take<char>(H * H, 2);
    if (L.cross_index >= 0) {
      L.wcqT = cv.take<char>(H * H, 2);
      L.wcoT = cv.take<char>(H * H, 2);
    }
    L.wiqT = cv.take<char>(I * H, 2);
    L.woqT = cv.take<char>(I * H, 2);
    L.witT = cv.take<char>(I * H, 2);
    L.wotT = cv.take<char>(I * H, 2);
  }
  return cv.off;
}

int refresh_transposes(mra_qformer* h, hipStream_t stream) {
  const mra_cfg& c = h->cfg;
  const int H = c.hidden, I = c.inter, op = h->op();
  if (!h->tr_jobs) {   // one launch for every weight: table of (matrix, first 32x32 tile)
    std::vector<TrJob> jobs;
    int tiles = 0;
    auto add = [&](const void* src, void* dst, int R, int C) {
      const int tx = (C + 31) / 32, ty = (R + 31) / 32;
      jobs.push_back(TrJob{src, dst, R, C, tiles, tx});
      tiles += tx * ty;
    };
    for (int i = 0; i < c.layers; ++i) {
      const LayerW& L = h->layers[i];
      add(L.wqkv, L.wqkvT, 3 * H, H);
      add(L.wo, L.woT, H, H);
      if (L.cross_index >= 0) {
        add(L.wcq, L.wcqT, H, H);
        add(L.wco, L.wcoT, H, H);
      }
      add(L.wiq, L.wiqT, I, H);
      add(L.woq, L.woqT, H, I);
      add(L.wit, L.witT, I, H);
      add(L.wot, L.wotT, H, I);
    }
    if (hipMalloc((void**)&h->tr_jobs, jobs.size() * sizeof(TrJob)) != hipSuccess) return -3;
    if (hipMemcpy(h->tr_jobs, jobs.data(), jobs.size() * sizeof(TrJob), hipMemcpyHostToDevice) != hipSuccess) return -4;
    h->n_tr_jobs = (int)jobs.size();
    h->n_tr_tiles = tiles;
  }
  const int rc = launch_transpose16_batch(h->tr_jobs, h->n_tr_jobs, h->n_tr_tiles, op, stream);
  if (rc) return rc;
  h->transposes_stale = false;
  return 0;
}

struct Ctx {
  mra_qformer* h;
  hipStream_t stream;
  int op;
  hipStream_t wstream;   // weight-gradient GEMMs: the handle's side stream (backward) or `stream`
  // forward-kernel GEMM:  C = A W^T (+bias) (+R); aux: the pre-activation tensor of EPI_GELU_BOTH / EPI_GELU_BWD
  int gemm(const void* A, RowView av, const void* W, const float* bias, void* C, RowView cv, const float* R, RowView rv, int M, int N,
           int K, int epi, void* aux = nullptr) const {
    if (M <= 0) return 0;
    GemmProb p{};
    p.A = A; p.a = av; p.W = W; p.bias = bias; p.C = C; p.c = cv; p.R = R; p.r = rv; p.aux = aux;
    p.M = M; p.N = N; p.K = K; p.tile_cfg = ring_tile(N, epi);
    return launch_gemm(&p, 1, epi, op, stream);
  }
  // mra_qformer_set_option("train_ring"): the ring kernel's exact-fit tiles where it has the epilogue (GemmProb::tile_cfg, 0 = automatic)
  int ring_tile(int N, int epi) const {
    const int mask = h->train_ring;
    if ((mask & 1) && epi == EPI_OP && N == 3 * h->cfg.hidden && N % 144 == 0) return 9;
    if ((mask & 4) && (epi == EPI_OP || epi == EPI_RES_F32) && N == h->cfg.hidden && N % 96 == 0) return 11;
    return 0;
  }
  static GemmProb prob(const void* A, RowView av, const void* W, const float* bias, void* C, RowView cv, const float* R, RowView rv, int M, int N,
                       int K, void* aux = nullptr) {
    GemmProb p{};
    p.A = A; p.a = av; p.W = W; p.bias = bias; p.C = C; p.c = cv; p.R = R; p.r = rv; p.aux = aux;
    p.M = M; p.N = N; p.K = K;
    return p;
  }
  // two problems with the same epilogue in one launch; an empty one (M <= 0) is dropped
  int gemm2(const GemmProb& p0, const GemmProb& p1, int epi) const {
    GemmProb ps[2];
    int n = 0;
    if (p0.M > 0) ps[n++] = p0;
    if (p1.M > 0) ps[n++] = p1;
    if (n) ps[0].tile_cfg = ring_tile(ps[0].N, epi);   // the first problem decides
    return n ? launch_gemm(ps, n, epi, op, stream) : 0;
  }
  // dW += dY^T X, db += colsum(dY); dY given as [M, ldy] row view starting at column block `cb0`.  Queued: the weight gradients that become
  // computable at the same point of the backward leave together in wflush() (launch_gemm_tn_group: up to four per launch)
  mutable std::vector<GemmTnArgs> pending;
  int wgrad(const void* dY, RowView yv, long long y_block_stride, const void* X, RowView xv, int M, int N, int K, float* dW, float* db) const {
    if (M <= 0) return 0;
    GemmTnArgs a{};
    a.dY = dY; a.X = X; a.dW = dW; a.yv = yv; a.xv = xv; a.y_block_stride = y_block_stride; a.x_block_stride = 64;
    a.M = M; a.N = N; a.K = K; a.ldw = K; a.accumulate = 1; a.db = db;
    pending.push_back(a);
    return 0;
  }
  int wflush() const {
    int rc = 0;
    for (size_t i = 0; i < pending.size() && !rc; i += GEMM_TN_MAX_JOBS)
      rc = launch_gemm_tn_group(pending.data() + i, (int)std::min<size_t>(GEMM_TN_MAX_JOBS, pending.size() - i), op, wstream);
    pending.clear();
    return rc;
  }
};

float* gptr(mra_qformer* h, float* grads, const std::string& name) {
  auto it = h->params.find(name);
  return it == h->params.end() ? nullptr : grads + it->second.goff / 4;
}

}  // namespace

extern "C" {

size_t mra_qformer_grad_bytes(mra_qformer* h) { return h ? h->grad_bytes : 0; }

int mra_qformer_grad_offset(mra_qformer* h, const char* name, size_t* offset_bytes, int64_t* numel) {
  if (!h || !name) return fail(MRA_EINVAL, "null argument");
  auto it = h->params.find(name);
  if (it == h->params.end()) return fail(MRA_ENAME, std::string("unknown parameter name: ") + name);
  if (offset_bytes) *offset_bytes = it->second.goff;
  if (numel) *numel = it->second.numel;
  return MRA_OK;
}

namespace {
// tables of the fused optimizer pass (see mra_handle.h)
int build_adam_tables(mra_qformer* h) {
  if (h->adam_jobs) return 0;
  const mra_cfg& c = h->cfg;
  const int H = c.hidden, I = c.inter;
  std::vector<AdamMatJob> jobs;
  std::map<std::string, bool> in_mats;
  int tiles = 0;
  auto add = [&](const std::string& name, void* dstT, int R, int C, int ldT) {
    const Param& pr = h->params.at(name);
    jobs.push_back(AdamMatJob{(unsigned long long)(pr.goff / 4), pr.ptr, dstT, R, C, ldT, tiles, C / 256});
    tiles += (R / 64) * (C / 256);
    in_mats[name] = true;
  };
  for (int i = 0; i < c.layers; ++i) {
    const LayerW& L = h->layers[i];
    const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
    const char* names[3] = {"query", "key", "value"};
    for (int j = 0; j < 3; ++j) add(p + "attention.self." + names[j] + ".weight", (char*)L.wqkvT + (size_t)j * H * 2, H, H, 3 * H);
    add(p + "attention.output.dense.weight", L.woT, H, H, H);
    if (L.cross_index >= 0) {
      add(p + "crossattention.self.query.weight", L.wcqT, H, H, H);
      add(p + "crossattention.output.dense.weight", L.wcoT, H, H, H);
    }
    add(p + "intermediate_query.dense.weight", L.wiqT, I, H, I);
    add(p + "output_query.dense.weight", L.woqT, H, I, H);
    add(p + "intermediate.dense.weight", L.witT, I, H, I);
    add(p + "output.dense.weight", L.wotT, H, I, H);
  }
  std::vector<FlatSeg> segs, c32;
  for (auto& kv : h->params) {
    const Param& pr = kv.second;
    const bool bert = kv.first.rfind("bert.", 0) == 0;
    if (!bert && kv.first != "query_tokens") continue;
    if (pr.copy32)
      for (long long o = 0; o < pr.numel; o += FLAT_SEG)
        c32.push_back(FlatSeg{(unsigned long long)(pr.goff / 4 + o), (char*)(pr.copy32 + o), (int)std::min<long long>(FLAT_SEG, pr.numel - o), MRA_F32});
    if (in_mats.count(kv.first)) continue;
    const size_t esz = pr.dtype == MRA_F32 ? 4 : 2;
    for (long long o = 0; o < pr.numel; o += FLAT_SEG)
      segs.push_back(FlatSeg{(unsigned long long)(pr.goff / 4 + o), (char*)pr.ptr + o * esz, (int)std::min<long long>(FLAT_SEG, pr.numel - o), pr.dtype});
  }
  auto up = [&](const void* src, size_t bytes, void** dst) {
    if (hipMalloc(dst, bytes ? bytes : 16) != hipSuccess) return -3;
    return bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess ? -4 : 0;
  };
  int rc = up(jobs.data(), jobs.size() * sizeof(AdamMatJob), (void**)&h->adam_jobs);
  if (!rc) rc = up(segs.data(), segs.size() * sizeof(FlatSeg), (void**)&h->adam_segs);
  if (!rc) rc = up(c32.data(), c32.size() * sizeof(FlatSeg), (void**)&h->c32_segs);
  h->n_adam_jobs = (int)jobs.size(); h->n_adam_tiles = tiles; h->n_adam_segs = (int)segs.size(); h->n_c32_segs = (int)c32.size();
  return rc;
}
}  // namespace

int mra_qformer_adam_step(mra_qformer* h, float* master, float* grad, float* exp_avg, float* exp_avg_sq, size_t bytes, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad, void* stream_) {
  if (!h || !master || !grad || !exp_avg || !exp_avg_sq) return fail(MRA_EINVAL, "null argument");
  if (!h->arena_t || !h->adam_jobs) return fail(MRA_ESTATE, "call mra_qformer_enable_training first");
  if (bytes < h->grad_bytes) return fail(MRA_EINVAL, "buffers smaller than mra_qformer_grad_bytes()");
  if (((size_t)master | (size_t)grad | (size_t)exp_avg | (size_t)exp_avg_sq) & 15) return fail(MRA_EINVAL, "buffers must be 16-byte aligned");
  if (step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return fail(MRA_EINVAL, "bad Adam hyper-parameters");
  if (h->cfg.hidden % 256 || h->cfg.inter % 256) return fail(MRA_EINVAL, "hidden / inter must be multiples of 256");
  hipStream_t st = as_stream(stream_);
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
  AdamScalars sc{beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)(1.0 / std::sqrt(bc2)), zero_grad ? 1 : 0};
  int rc = launch_adam_mats(master, grad, exp_avg, exp_avg_sq, h->adam_jobs, h->n_adam_jobs, h->n_adam_tiles, sc, h->op(), st);
  if (!rc) rc = launch_adam_flat(master, grad, exp_avg, exp_avg_sq, h->adam_segs, h->n_adam_segs, sc, st);
  if (!rc) rc = launch_convert_flat(master, h->c32_segs, h->n_c32_segs, st);
  if (rc) return chk(rc, "fused Adam pass");
  for (auto& kv : h->params)
    if (kv.first.rfind("bert.", 0) == 0) kv.second.loaded = true;
  h->transposes_stale = false;   // the pass wrote the transposed copies too
  h->fold_stale = true;
  h->precise_stale = true;
  return MRA_OK;
}

int mra_qformer_enable_training(mra_qformer* h, void* stream) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (!h->arena_t) {
    const size_t bytes = layout_transposes(h, nullptr);
    hipError_t e = hipMalloc((void**)&h->arena_t, bytes);
    if (e != hipSuccess) return fail(MRA_ENOMEM, std::string("hipMalloc of transposed weights: ") + hipGetErrorString(e));
    layout_transposes(h, h->arena_t);
    h->transposes_stale = true;
  }
  if (!h->wg_stream) {   // weight-gradient GEMMs run beside the data-gradient chain (mra_qformer_backward)
    HIP_TRY(hipStreamCreateWithFlags(&h->wg_stream, hipStreamNonBlocking));
    for (auto& e : h->wg_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (build_adam_tables(h)) return fail(MRA_ENOMEM, "tables of the fused optimizer pass");
  if (h->transposes_stale) return chk(refresh_transposes(h, as_stream(stream)), "weight transposes");
  return MRA_OK;
}

size_t mra_qformer_train_workspace_bytes(mra_qformer* h, int32_t items, int32_t L, int32_t kv) {
  if (!h || items <= 0 || L < 0 || kv <= 0) return 0;
  return layout_train(h, nullptr, items, L, kv).bytes;
}

int mra_qformer_forward_train(mra_qformer* h, const int64_t* input_ids, const int64_t* attention_mask, const void* enc,
                              int32_t items, int32_t L, int32_t kv, float* out_query, float* out_cls, void* workspace,
                              size_t workspace_bytes, void* stream_) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (items <= 0 || L < 0 || kv <= 0) return fail(MRA_EINVAL, "bad sizes");
  const mra_cfg& c = h->cfg;
  if (L > c.max_pos || !enc || (L > 0 && !input_ids) || (out_cls && L < 1)) return fail(MRA_EINVAL, "bad arguments");
  if (!workspace || workspace_bytes < mra_qformer_train_workspace_bytes(h, items, L, kv) || reinterpret_cast<uintptr_t>(workspace) % 256)
    return fail(MRA_ENOMEM, "training workspace too small or misaligned");
  hipStream_t stream = as_stream(stream_);
  const int N = items, Q = c.n_query, S = Q + L, H = c.hidden, I = c.inter;
  const Ctx X{h, stream, h->op(), stream, {}};
  const int op = X.op;
  TrainBufs t = layout_train(h, (char*)workspace, N, L, kv);
  const long long SH = (long long)S * H;
  const RowView all_rows = plain(N * S, H), q_view = items_view(SH, Q, H), t_view = items_view(SH, L > 0 ? L : 1, H);
  const RowView qc_rows = plain(N * Q, H);
  const size_t t16 = (size_t)Q * H * 2, t32 = (size_t)Q * H;
  int rc = launch_embed_ln((const long long*)input_ids, N, L, Q, H, c.vocab, h->query, 0, h->word, h->pos, h->embg, h->embb, c.ln_eps,
                           t.layer[0].hin32, t.layer[0].hin16, t.emb_pre, op, stream);
  if (rc) return chk(rc, "embed_ln");
  if (h->ncross > 0 && (rc = kv_project(h, enc, N, kv, t.kv16, stream))) return chk(rc, "kv projection");
  for (int i = 0; i < c.layers; ++i) {
    const LayerW& W = h->layers[i];
    LayerBuf& b = t.layer[i];
    float* nxt32 = i + 1 < c.layers ? t.layer[i + 1].hin32 : t.out32;
    char* nxt16 = i + 1 < c.layers ? t.layer[i + 1].hin16 : t.out16;
    if ((rc = X.gemm(b.hin16, all_rows, W.wqkv, W.bqkv, b.qkv16, plain(N * S, 3 * H), nullptr, all_rows, N * S, 3 * H, H, EPI_OP)))
      return chk(rc, "qkv gemm");
    {
      AttnArgs a{};
      a.Q = b.qkv16; a.K = b.qkv16 + (size_t)H * 2; a.V = b.qkv16 + (size_t)2 * H * 2; a.O = b.ctx16;
      a.q_item_stride = a.k_item_stride = a.v_item_stride = (long long)S * 3 * H; a.q_ld = a.k_ld = a.v_ld = 3 * H;
      a.k_head_stride = a.v_head_stride = 64; a.o_item_stride = SH; a.o_ld = H;
      a.mask = (const long long*)attention_mask; a.mask_ld = S;
      a.items = N; a.heads = c.heads; a.q_rows = S; a.kv_len = S; a.scale = 0.125f; a.nsplit = 1; a.lse = b.lse_s;
      if ((rc = launch_attention(a, op, stream))) return chk(rc, "self attention");
    }
    if ((rc = X.gemm(b.ctx16, all_rows, W.wo, W.bo, b.pre1, all_rows, b.hin32, all_rows, N * S, H, H, EPI_RES_F32))) return chk(rc, "attn out gemm");
    if ((rc = launch_ln_rows(b.pre1, all_rows, N * S, H, W.ln1g, W.ln1b, c.ln_eps, b.h1_32, all_rows, b.h1_16, all_rows, op, stream)))
      return chk(rc, "attn ln");
    const void* fq16 = b.h1_16; const float* fq32 = b.h1_32; RowView fqv = q_view;
    if (W.cross_index >= 0) {
      if ((rc = X.gemm(b.h1_16, q_view, W.wcq, W.bcq, b.qc16, qc_rows, nullptr, qc_rows, N * Q, H, H, EPI_OP))) return chk(rc, "cross q gemm");
      AttnArgs a{};
      const size_t per_sel = (size_t)N * c.heads * kv * 64;
      a.Q = b.qc16; a.K = t.kv16 + (size_t)(W.cross_index * 2) * per_sel * 2; a.V = t.kv16 + (size_t)(W.cross_index * 2 + 1) * per_sel * 2;
      a.O = b.cctx16; a.q_item_stride = a.o_item_stride = (long long)Q * H; a.q_ld = a.o_ld = H;
      a.k_item_stride = a.v_item_stride = (long long)c.heads * kv * 64; a.k_head_stride = a.v_head_stride = (long long)kv * 64; a.k_ld = a.v_ld = 64;
      a.items = N; a.heads = c.heads; a.q_rows = Q; a.kv_len = kv; a.scale = 0.125f; a.nsplit = t.nsplit; a.part = t.part; a.lse = b.lse_c;
      if ((rc = launch_attention(a, op, stream))) return chk(rc, "cross attention");
      if ((rc = X.gemm(b.cctx16, qc_rows, W.wco, W.bco, b.pre2, qc_rows, b.h1_32, q_view, N * Q, H, H, EPI_RES_F32))) return chk(rc, "cross out gemm");
      if ((rc = launch_ln_rows(b.pre2, qc_rows, N * Q, H, W.lncg, W.lncb, c.ln_eps, b.hc32, qc_rows, b.hc16, qc_rows, op, stream))) return chk(rc, "cross ln");
      fq16 = b.hc16; fq32 = b.hc32; fqv = qc_rows;
    }
    // feed-forwards: rows [0, N*Q) of u16 / f16 are the query rows, [N*Q, N*S) the text rows
    char* u_t = b.u16 + (size_t)N * Q * I * 2;
    char* f_t = b.f16 + (size_t)N * Q * I * 2;
    // up-projection + GELU in one epilogue: u16 keeps the pre-activation for the backward, f16 is the activation
    // the query and the text feed-forward as two problems of one launch each (as the inference forward does), their LayerNorms as one launch
    if ((rc = X.gemm2(X.prob(fq16, fqv, W.wiq, W.biq, b.f16, plain(N * Q, I), nullptr, qc_rows, N * Q, I, H, b.u16),
                      X.prob(b.h1_16 + t16, t_view, W.wit, W.bit, f_t, plain(N * L, I), nullptr, qc_rows, N * L, I, H, u_t), EPI_GELU_BOTH))) return chk(rc, "ffn up");
    if ((rc = X.gemm2(X.prob(b.f16, plain(N * Q, I), W.woq, W.boq, b.pre3, q_view, fq32, fqv, N * Q, H, I),
                      X.prob(f_t, plain(N * L, I), W.wot, W.bot, b.pre3 + t32, t_view, b.h1_32 + t32, t_view, N * L, H, I), EPI_RES_F32))) return chk(rc, "ffn down");
    if (L > 0) {
      if ((rc = launch_ln_rows2(b.pre3, all_rows, N * S, H, W.lnqg, W.lnqb, W.lntg, W.lntb, S, Q, c.ln_eps, nxt32, all_rows, nxt16, all_rows, op, stream))) return chk(rc, "ffn ln");
    } else if ((rc = launch_ln_rows(b.pre3, q_view, N * Q, H, W.lnqg, W.lnqb, c.ln_eps, nxt32, q_view, nxt16, q_view, op, stream))) return chk(rc, "ffn-q ln");
  }
  if (out_query && (rc = launch_copy_rows_f32(t.out32, q_view, out_query, qc_rows, N * Q, H, stream))) return chk(rc, "copy out_query");
  if (out_cls && (rc = launch_copy_rows_f32(t.out32 + t32, items_view(SH, 1, H), out_cls, plain(N, H), N, H, stream))) return chk(rc, "copy out_cls");
  return MRA_OK;
}

int mra_qformer_backward(mra_qformer* h, const int64_t* input_ids, const int64_t* attention_mask, const void* enc, int32_t items,
                         int32_t L, int32_t kv, const float* d_out_query, const float* d_out_cls, float* grads, void* workspace,
                         size_t workspace_bytes, void* stream_) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (items <= 0 || L < 0 || kv <= 0 || !grads || !enc) return fail(MRA_EINVAL, "bad arguments");
  if (!d_out_query && !d_out_cls) return fail(MRA_EINVAL, "no upstream gradient");
  if (!h->arena_t || h->transposes_stale) return fail(MRA_ESTATE, "call mra_qformer_enable_training after the last weight upload");
  if (!workspace || workspace_bytes < mra_qformer_train_workspace_bytes(h, items, L, kv)) return fail(MRA_ENOMEM, "training workspace too small");
  const mra_cfg& c = h->cfg;
  hipStream_t stream = as_stream(stream_);
  const int N = items, Q = c.n_query, S = Q + L, H = c.hidden, I = c.inter;
  if (!h->wg_stream) return fail(MRA_ESTATE, "call mra_qformer_enable_training first");
  // Weight gradients are off the critical path (nothing in this call reads them): they run on the handle's side stream beside
  // the data-gradient chain.  fork(): the side stream waits for everything issued on `stream` so far (the dY it is about to
  // read); done(): an event on the side stream that `stream` waits for before it overwrites a scratch tensor those GEMMs read;
  // the call ends with a join, so to the caller everything still happens on `stream`.
  const Ctx X{h, stream, h->op(), h->wg_stream, {}};   // measured: backward 10.7 -> 8.6 ms at B = 1 x T = 20, both modalities
  const int op = X.op;
  int evi = 0;
  auto next_event = [&]() { return h->wg_ev[evi++ % (int)(sizeof(h->wg_ev) / sizeof(h->wg_ev[0]))]; };
  // A failed record / wait would turn into an unsynchronised read of a scratch tensor or a missing join: the first HIP error is kept
  // and reported, and EVERY exit after the first fork -- error exits included -- joins the side stream before returning (the caller
  // may reuse or free the workspace and gradient buffers right after the call).
  hipError_t sync_err = hipSuccess;
  bool forked = false;
  auto note = [&](hipError_t r) { if (r != hipSuccess && sync_err == hipSuccess) sync_err = r; };
  auto fork = [&]() {
    hipEvent_t e = next_event();
    forked = true;
    hipError_t r = hipEventRecord(e, stream);
    if (r == hipSuccess) r = hipStreamWaitEvent(h->wg_stream, e, 0);
    note(r);
  };
  auto done = [&]() { hipEvent_t e = next_event(); note(hipEventRecord(e, h->wg_stream)); return e; };
  auto wait_for = [&](hipEvent_t e) { note(hipStreamWaitEvent(stream, e, 0)); };
  auto body = [&]() -> int {
  hipEvent_t ffn_done = nullptr, attn_done = nullptr;   // weight gradients of the previously processed layer
  TrainBufs t = layout_train(h, (char*)workspace, N, L, kv);
  const long long SH = (long long)S * H;
  const RowView all_rows = plain(N * S, H), q_view = items_view(SH, Q, H), t_view = items_view(SH, L > 0 ? L : 1, H);
  const RowView qc_rows = plain(N * Q, H);
  const size_t t16 = (size_t)Q * H * 2, t32 = (size_t)Q * H;
  int rc;
  // upstream gradient of last_hidden_state: zeros, query rows <- d_out_query, [CLS] row <- d_out_cls
  float* dh = t.dhA;     // gradient w.r.t. the current layer's output
  float* dh1 = t.dhB;    // gradient w.r.t. the post-self-attention state h1
  HIP_TRY(hipMemsetAsync(dh, 0, (size_t)N * S * H * 4, stream));
  if (d_out_query && (rc = launch_copy_rows_f32(d_out_query, qc_rows, dh, q_view, N * Q, H, stream))) return chk(rc, "seed dq");
  if (d_out_cls && (rc = launch_copy_rows_f32(d_out_cls, plain(N, H), dh + t32, items_view(SH, 1, H), N, H, stream))) return chk(rc, "seed dcls");
  if (h->ncross > 0) HIP_TRY(hipMemsetAsync(t.dkv16, 0, (size_t)h->ncross * 2 * N * kv * H * 2, stream));
  auto G = [&](const std::string& n) { return gptr(h, grads, n); };

  for (int i = c.layers - 1; i >= 0; --i) {
    const LayerW& W = h->layers[i];
    const LayerBuf& b = t.layer[i];
    const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
    const void* fq16 = W.cross_index >= 0 ? (const void*)b.hc16 : (const void*)b.h1_16;
    const RowView fqv = W.cross_index >= 0 ? qc_rows : q_view;
    const char* u_t = b.u16 + (size_t)N * Q * I * 2;
    const char* f_t = b.f16 + (size_t)N * Q * I * 2;
    char* dff_t = t.dff16 + (size_t)N * Q * I * 2;
    if (ffn_done) wait_for(ffn_done);   // dpre16 / dff16 / dpre2_16 / dqc16 are about to be rewritten
    // ---- feed-forwards, query and text rows together: out = LN(pre3), pre3 = f Wo^T + b + input; each step is ONE launch over both branches ----
    const char* dpt16 = t.dpre16 + (size_t)N * Q * H * 2;
    const float* dpt32 = t.dpre32 + (size_t)N * Q * H;
    {
      LnBwdArgs a{}, at{};
      a.dy = dh; a.dyv = q_view; a.x = b.pre3; a.xv = q_view; a.gamma = W.lnqg; a.eps = c.ln_eps; a.rows = N * Q;
      a.dx = t.dpre32; a.dxv = qc_rows; a.dx16 = t.dpre16; a.dx16v = qc_rows;
      a.dgamma = G(p + "output_query.LayerNorm.weight"); a.dbeta = G(p + "output_query.LayerNorm.bias");
      at.dy = dh + t32; at.dyv = t_view; at.x = b.pre3 + t32; at.xv = t_view; at.gamma = W.lntg; at.eps = c.ln_eps; at.rows = N * L;
      at.dx = t.dpre32 + (size_t)N * Q * H; at.dxv = plain(N * L, H); at.dx16 = t.dpre16 + (size_t)N * Q * H * 2; at.dx16v = plain(N * L, H);
      at.dgamma = G(p + "output.LayerNorm.weight"); at.dbeta = G(p + "output.LayerNorm.bias");
      if ((rc = launch_ln_bwd2(a, L > 0 ? &at : nullptr, H, op, stream))) return chk(rc, "ffn ln bwd");
    }
    // d(pre-activation) = (d_pre3 Wo) * gelu'(u): the GELU gradient is the data-gradient GEMM's epilogue
    if ((rc = X.gemm2(X.prob(t.dpre16, qc_rows, W.woqT, nullptr, t.dff16, plain(N * Q, I), nullptr, qc_rows, N * Q, I, H, b.u16),
                      X.prob(dpt16, plain(N * L, H), W.wotT, nullptr, dff_t, plain(N * L, I), nullptr, qc_rows, N * L, I, H, (void*)u_t), EPI_GELU_BWD))) return chk(rc, "dff");
    fork();
    if ((rc = X.wgrad(t.dpre16, qc_rows, 64, b.f16, plain(N * Q, I), N * Q, H, I, G(p + "output_query.dense.weight"), G(p + "output_query.dense.bias"))))
      return chk(rc, "dWoq");
    if ((rc = X.wgrad(t.dff16, plain(N * Q, I), 64, fq16, fqv, N * Q, I, H, G(p + "intermediate_query.dense.weight"), G(p + "intermediate_query.dense.bias"))))
      return chk(rc, "dWiq");
    if (L > 0) {
      if ((rc = X.wgrad(dpt16, plain(N * L, H), 64, f_t, plain(N * L, I), N * L, H, I, G(p + "output.dense.weight"), G(p + "output.dense.bias")))) return chk(rc, "dWot");
      if ((rc = X.wgrad(dff_t, plain(N * L, I), 64, b.h1_16 + t16, t_view, N * L, I, H, G(p + "intermediate.dense.weight"), G(p + "intermediate.dense.bias"))))
        return chk(rc, "dWit");
    }
    // d(input) = d_pre3 (residual) + du Wi: query rows -> compact dhc32 (cross layers) or the query rows of dh1; text rows -> dh1
    float* dfq = W.cross_index >= 0 ? t.dhc32 : dh1;
    const RowView dfqv = W.cross_index >= 0 ? qc_rows : q_view;
    if ((rc = X.gemm2(X.prob(t.dff16, plain(N * Q, I), W.wiqT, nullptr, dfq, dfqv, t.dpre32, qc_rows, N * Q, H, I),
                      X.prob(dff_t, plain(N * L, I), W.witT, nullptr, dh1 + t32, t_view, dpt32, plain(N * L, H), N * L, H, I), EPI_RES_F32))) return chk(rc, "d_ffn_in");
    // ---- cross-attention block: hc = LN(pre2), pre2 = cctx Wco^T + b + h1[:, :32] ----
    if (W.cross_index >= 0) {
      LnBwdArgs a{};
      a.dy = t.dhc32; a.dyv = qc_rows; a.x = b.pre2; a.xv = qc_rows; a.gamma = W.lncg; a.eps = c.ln_eps; a.rows = N * Q;
      a.dx = t.dpre2_32; a.dxv = qc_rows; a.dx16 = t.dpre2_16; a.dx16v = qc_rows;
      a.dgamma = G(p + "crossattention.output.LayerNorm.weight"); a.dbeta = G(p + "crossattention.output.LayerNorm.bias");
      if ((rc = launch_ln_bwd(a, H, op, stream))) return chk(rc, "cross ln bwd");
      if ((rc = X.gemm(t.dpre2_16, qc_rows, W.wcoT, nullptr, t.dcctx16, qc_rows, nullptr, qc_rows, N * Q, H, H, EPI_OP))) return chk(rc, "d_cctx");
      AttnBwdArgs g{};
      const size_t per_sel = (size_t)N * c.heads * kv * 64;
      g.Q = b.qc16; g.K = t.kv16 + (size_t)(W.cross_index * 2) * per_sel * 2; g.V = t.kv16 + (size_t)(W.cross_index * 2 + 1) * per_sel * 2;
      g.O = b.cctx16; g.dO = t.dcctx16;
      g.dQ = t.dqc16; g.dK = t.dkv16 + (size_t)(W.cross_index * 2) * per_sel * 2; g.dV = t.dkv16 + (size_t)(W.cross_index * 2 + 1) * per_sel * 2;
      g.q_item_stride = g.o_item_stride = g.dq_item_stride = (long long)Q * H; g.q_ld = g.o_ld = g.dq_ld = H;
      g.k_item_stride = g.v_item_stride = g.dk_item_stride = g.dv_item_stride = (long long)c.heads * kv * 64;
      g.k_head_stride = g.v_head_stride = g.dk_head_stride = g.dv_head_stride = (long long)kv * 64;
      g.k_ld = g.v_ld = g.dk_ld = g.dv_ld = 64;
      g.lse = b.lse_c; g.items = N; g.heads = c.heads; g.q_rows = Q; g.kv_len = kv; g.scale = 0.125f;
      if ((rc = launch_attn_bwd(g, op, stream))) return chk(rc, "cross attention bwd");
      fork();
      if ((rc = X.wgrad(t.dpre2_16, qc_rows, 64, b.cctx16, qc_rows, N * Q, H, H, G(p + "crossattention.output.dense.weight"), G(p + "crossattention.output.dense.bias"))))
        return chk(rc, "dWco");
      if ((rc = X.wgrad(t.dqc16, qc_rows, 64, b.h1_16, q_view, N * Q, H, H, G(p + "crossattention.self.query.weight"), G(p + "crossattention.self.query.bias"))))
        return chk(rc, "dWcq");
      if ((rc = X.gemm(t.dqc16, qc_rows, W.wcqT, nullptr, dh1, q_view, t.dpre2_32, qc_rows, N * Q, H, H, EPI_RES_F32))) return chk(rc, "d_h1q");
    }
    if ((rc = X.wflush())) return chk(rc, "feed-forward / cross-attention weight gradients");
    ffn_done = done();   // behind this layer's feed-forward and cross-attention weight gradients
    // ---- self-attention block: h1 = LN(pre1), pre1 = ctx Wo^T + b + hin ----
    if (attn_done) wait_for(attn_done);   // dpre16b / dqkv16 are about to be rewritten
    {
      LnBwdArgs a{};
      a.dy = dh1; a.dyv = all_rows; a.x = b.pre1; a.xv = all_rows; a.gamma = W.ln1g; a.eps = c.ln_eps; a.rows = N * S;
      a.dx = t.dpre32; a.dxv = all_rows; a.dx16 = t.dpre16b; a.dx16v = all_rows;
      a.dgamma = G(p + "attention.output.LayerNorm.weight"); a.dbeta = G(p + "attention.output.LayerNorm.bias");
      if ((rc = launch_ln_bwd(a, H, op, stream))) return chk(rc, "attn ln bwd");
    }
    if ((rc = X.gemm(t.dpre16b, all_rows, W.woT, nullptr, t.dctx16, all_rows, nullptr, all_rows, N * S, H, H, EPI_OP))) return chk(rc, "d_ctx");
    {
      AttnBwdArgs g{};
      g.Q = b.qkv16; g.K = b.qkv16 + (size_t)H * 2; g.V = b.qkv16 + (size_t)2 * H * 2; g.O = b.ctx16; g.dO = t.dctx16;
      g.dQ = t.dqkv16; g.dK = t.dqkv16 + (size_t)H * 2; g.dV = t.dqkv16 + (size_t)2 * H * 2;
      g.q_item_stride = g.k_item_stride = g.v_item_stride = g.dq_item_stride = g.dk_item_stride = g.dv_item_stride = (long long)S * 3 * H;
      g.q_ld = g.k_ld = g.v_ld = g.dq_ld = g.dk_ld = g.dv_ld = 3 * H;
      g.k_head_stride = g.v_head_stride = g.dk_head_stride = g.dv_head_stride = 64;
      g.o_item_stride = SH; g.o_ld = H;
      g.mask = (const long long*)attention_mask; g.mask_ld = S; g.lse = b.lse_s;
      g.items = N; g.heads = c.heads; g.q_rows = S; g.kv_len = S; g.scale = 0.125f;
      if ((rc = launch_attn_bwd(g, op, stream))) return chk(rc, "self attention bwd");
    }
    fork();
    if ((rc = X.wgrad(t.dpre16b, all_rows, 64, b.ctx16, all_rows, N * S, H, H, G(p + "attention.output.dense.weight"), G(p + "attention.output.dense.bias"))))
      return chk(rc, "dWo");
    const char* names[3] = {"query", "key", "value"};
    for (int j = 0; j < 3; ++j)
      if ((rc = X.wgrad(t.dqkv16 + (size_t)j * H * 2, plain(N * S, 3 * H), 64, b.hin16, all_rows, N * S, H, H,
                        G(p + "attention.self." + names[j] + ".weight"), G(p + "attention.self." + names[j] + ".bias"))))
        return chk(rc, "dWqkv");
    if ((rc = X.wflush())) return chk(rc, "self-attention weight gradients");
    attn_done = done();
    // gradient w.r.t. the layer input: d_pre1 (residual) + dqkv Wqkv  -> becomes dh of layer i-1
    if ((rc = X.gemm(t.dqkv16, plain(N * S, 3 * H), W.wqkvT, nullptr, dh, all_rows, t.dpre32, all_rows, N * S, H, 3 * H, EPI_RES_F32))) return chk(rc, "d_hin");
  }
  // ---- embeddings: h0 = LN(emb_pre) ----
  {
    LnBwdArgs a{};
    a.dy = dh; a.dyv = all_rows; a.x = t.emb_pre; a.xv = all_rows; a.gamma = h->embg; a.eps = c.ln_eps; a.rows = N * S;
    a.dx = t.dpre32; a.dxv = all_rows;
    a.dgamma = G("bert.embeddings.LayerNorm.weight"); a.dbeta = G("bert.embeddings.LayerNorm.bias");
    if ((rc = launch_ln_bwd(a, H, op, stream))) return chk(rc, "embedding ln bwd");
    if ((rc = launch_embed_bwd(t.dpre32, (const long long*)input_ids, N, L, Q, H, c.vocab, G("query_tokens"), G("bert.embeddings.position_embeddings.weight"),
                               G("bert.embeddings.word_embeddings.weight"), stream)))
      return chk(rc, "embedding bwd");
  }
  // ---- cross K/V projections: dWk / dWv of every cross layer from the head-major dK / dV cache ----
  fork();
  for (int i = 0; i < c.layers; ++i) {
    const int cl = h->layers[i].cross_index;
    if (cl < 0) continue;
    const std::string p = "bert.encoder.layer." + std::to_string(i) + ".crossattention.self.";
    const size_t per_sel = (size_t)N * c.heads * kv * 64;
    const RowView hv = items_view((long long)c.heads * kv * 64, kv, 64);
    for (int kvsel = 0; kvsel < 2; ++kvsel) {
      const char* dY = t.dkv16 + (size_t)(cl * 2 + kvsel) * per_sel * 2;
      const std::string nm = p + (kvsel ? "value" : "key");
      if ((rc = X.wgrad(dY, hv, (long long)kv * 64, enc, plain(N * kv, c.enc_width), N * kv, H, c.enc_width, G(nm + ".weight"), G(nm + ".bias"))))
        return chk(rc, "dWkv");
    }
  }
  if ((rc = X.wflush())) return chk(rc, "cross K / V weight gradients");
  return MRA_OK;
  };   // body
  const int rc_body = body();
  if (forked) {   // join: what follows on `stream` sees every gradient, and nothing on the side stream still reads the caller's buffers
    hipEvent_t e = next_event();
    hipError_t r = hipEventRecord(e, h->wg_stream);
    if (r == hipSuccess) r = hipStreamWaitEvent(stream, e, 0);
    if (r != hipSuccess) { note(r); (void)hipStreamSynchronize(h->wg_stream); }   // last resort: a host-side join
  }
  if (rc_body) return rc_body;
  if (sync_err != hipSuccess) return fail(MRA_EHIP, std::string("side-stream synchronisation of the weight gradients: ") + hipGetErrorString(sync_err));
  return MRA_OK;
}

}  // extern "C"
