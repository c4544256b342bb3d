// C ABI (include/mra.h) and the host-side orchestration of the Q-Former forward.
//
// Mirrors, launch by launch, what the reference gets from LAVIS' BertModel when it calls
// {modality}_Qformer.bert(...) at models/xinstructblip.py:286-293 (layer structure as stated by
// the HF port, modeling_instructblip.py:591-709), re-planned for gfx950:
//   * the K/V projections of all cross-attention layers read the same encoder features, so they
//     run as ONE GEMM [items*kv, E] x [n_cross*2*H, E]^T whose epilogue scatters into a head-major
//     K/V cache ([kv][64] contiguous per (layer, k|v, item, head)) -- the layout the attention
//     kernel streams with full 128-byte rows;
//   * query rows and text rows of the [items, 32+L, H] residual stream are addressed in place
//     through row views (no slicing copies); the two feed-forwards run as one grouped launch;
//   * residual stream, LayerNorm statistics and softmax are fp32; only MFMA operands are f16/bf16.
// No allocation and no synchronisation after create/load; everything is enqueued on `stream`.
#include <cstdio>
#include <cstdlib>

#include "mra_handle.h"

using namespace mra;
using namespace mra_host;

namespace mra_host {
thread_local std::string g_err;
}

namespace {

// Lays the parameter arena out; with base == nullptr only measures.
size_t layout_params(mra_qformer* h, char* base) {
  const mra_cfg& c = h->cfg;
  const size_t H = c.hidden, I = c.inter, E = c.enc_width;
  Carver cv(base);
  size_t goff = 0;
  auto reg = [&](const std::string& name, void* p, int dtype, long long numel) {
    Param pr;
    pr.ptr = p;
    pr.dtype = dtype;
    pr.numel = numel;
    pr.goff = goff;  // gradients: one f32 block per parameter, registration order
    goff += align_up((size_t)numel * 4);
    h->params[name] = pr;
  };
  const int opd = c.op_dtype;
  h->params.clear();
  h->layers.assign(c.layers, LayerW{});
  h->word = cv.take<float>((size_t)c.vocab * H);
  h->pos = cv.take<float>((size_t)c.max_pos * H);
  h->embg = cv.take<float>(H);
  h->embb = cv.take<float>(H);
  h->query = cv.take<float>((size_t)c.n_query * H);
  h->encg = cv.take<float>(E);
  h->encb = cv.take<float>(E);
  reg("bert.embeddings.word_embeddings.weight", h->word, MRA_F32, (long long)c.vocab * H);
  reg("bert.embeddings.position_embeddings.weight", h->pos, MRA_F32, (long long)c.max_pos * H);
  reg("bert.embeddings.LayerNorm.weight", h->embg, MRA_F32, H);
  reg("bert.embeddings.LayerNorm.bias", h->embb, MRA_F32, H);
  reg("query_tokens", h->query, MRA_F32, (long long)c.n_query * H);
  reg("ln.weight", h->encg, MRA_F32, E);
  reg("ln.bias", h->encb, MRA_F32, E);
  h->ncross = 0;
  for (int i = 0; i < c.layers; ++i)
    if (i % c.cross_freq == 0) ++h->ncross;
  h->wkv = cv.take<char>((size_t)h->ncross * 2 * H * E, 2);
  h->bkv = cv.take<float>((size_t)h->ncross * 2 * H);
  h->wk32 = cv.take<float>((size_t)h->ncross * H * E);
  if (c.llm_hidden > 0) {
    h->wllm = cv.take<char>((size_t)c.llm_hidden * H, 2);
    h->bllm = cv.take<float>(c.llm_hidden);
    reg("llm_proj.weight", h->wllm, opd, (long long)c.llm_hidden * H);
    reg("llm_proj.bias", h->bllm, MRA_F32, c.llm_hidden);
  }
  int cl = 0;
  for (int i = 0; i < c.layers; ++i) {
    LayerW& L = h->layers[i];
    const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
    char* wqkv = cv.take<char>(3 * H * H, 2);
    L.wqkv = wqkv;
    L.bqkv = cv.take<float>(3 * H);
    const char* names[3] = {"query", "key", "value"};
    for (int j = 0; j < 3; ++j) {
      reg(p + "attention.self." + names[j] + ".weight", wqkv ? wqkv + (size_t)j * H * H * 2 : nullptr, opd, H * H);
      reg(p + "attention.self." + names[j] + ".bias", L.bqkv ? L.bqkv + j * H : nullptr, MRA_F32, H);
    }
    L.wo = cv.take<char>(H * H, 2);
    L.bo = cv.take<float>(H);
    L.ln1g = cv.take<float>(H);
    L.ln1b = cv.take<float>(H);
    reg(p + "attention.output.dense.weight", L.wo, opd, H * H);
    reg(p + "attention.output.dense.bias", L.bo, MRA_F32, H);
    reg(p + "attention.output.LayerNorm.weight", L.ln1g, MRA_F32, H);
    reg(p + "attention.output.LayerNorm.bias", L.ln1b, MRA_F32, H);
    L.cross_index = -1;
    if (i % c.cross_freq == 0) {
      L.cross_index = cl;
      L.wcq = cv.take<char>(H * H, 2);
      L.wcq32 = cv.take<float>(H * H);
      L.bcq = cv.take<float>(H);
      L.wco = cv.take<char>(H * H, 2);
      L.bco = cv.take<float>(H);
      L.lncg = cv.take<float>(H);
      L.lncb = cv.take<float>(H);
      reg(p + "crossattention.self.query.weight", L.wcq, opd, H * H);
      h->params[p + "crossattention.self.query.weight"].copy32 = L.wcq32;
      reg(p + "crossattention.self.query.bias", L.bcq, MRA_F32, H);
      char* wkv = (char*)h->wkv;
      reg(p + "crossattention.self.key.weight", wkv ? wkv + (size_t)(cl * 2 + 0) * H * E * 2 : nullptr, opd, H * E);
      h->params[p + "crossattention.self.key.weight"].copy32 = h->wk32 ? h->wk32 + (size_t)cl * H * E : nullptr;
      reg(p + "crossattention.self.key.bias", h->bkv ? h->bkv + (size_t)(cl * 2 + 0) * H : nullptr, MRA_F32, H);
      reg(p + "crossattention.self.value.weight", wkv ? wkv + (size_t)(cl * 2 + 1) * H * E * 2 : nullptr, opd, H * E);
      reg(p + "crossattention.self.value.bias", h->bkv ? h->bkv + (size_t)(cl * 2 + 1) * H : nullptr, MRA_F32, H);
      reg(p + "crossattention.output.dense.weight", L.wco, opd, H * H);
      reg(p + "crossattention.output.dense.bias", L.bco, MRA_F32, H);
      reg(p + "crossattention.output.LayerNorm.weight", L.lncg, MRA_F32, H);
      reg(p + "crossattention.output.LayerNorm.bias", L.lncb, MRA_F32, H);
      ++cl;
    }
    L.wit = cv.take<char>(I * H, 2);
    L.bit = cv.take<float>(I);
    L.wot = cv.take<char>(H * I, 2);
    L.bot = cv.take<float>(H);
    L.lntg = cv.take<float>(H);
    L.lntb = cv.take<float>(H);
    L.wiq = cv.take<char>(I * H, 2);
    L.biq = cv.take<float>(I);
    L.woq = cv.take<char>(H * I, 2);
    L.boq = cv.take<float>(H);
    L.lnqg = cv.take<float>(H);
    L.lnqb = cv.take<float>(H);
    reg(p + "intermediate.dense.weight", L.wit, opd, I * H);
    reg(p + "intermediate.dense.bias", L.bit, MRA_F32, I);
    reg(p + "output.dense.weight", L.wot, opd, H * I);
    reg(p + "output.dense.bias", L.bot, MRA_F32, H);
    reg(p + "output.LayerNorm.weight", L.lntg, MRA_F32, H);
    reg(p + "output.LayerNorm.bias", L.lntb, MRA_F32, H);
    reg(p + "intermediate_query.dense.weight", L.wiq, opd, I * H);
    reg(p + "intermediate_query.dense.bias", L.biq, MRA_F32, I);
    reg(p + "output_query.dense.weight", L.woq, opd, H * I);
    reg(p + "output_query.dense.bias", L.boq, MRA_F32, H);
    reg(p + "output_query.LayerNorm.weight", L.lnqg, MRA_F32, H);
    reg(p + "output_query.LayerNorm.bias", L.lnqb, MRA_F32, H);
  }
  h->grad_bytes = goff;
  return cv.off;
}

// Workspace of one forward; with base == nullptr only measures.
struct Work {
  float *hA32, *hB32, *pre32, *hC32, *part;
  char *hA16, *hB16, *qkv16, *ctx16, *qc16, *hC16, *ffn16, *kv16;
  char *encT, *qp16, *p16, *u16;   // folded cross-attention: enc^T [N][E][kvp], Q' [N][R][E], P [N][R][kvp], U [N][R][E]
  char *hs16, *qs16;               // split-precision cross-attention: (hi | lo | hi) of the query rows [N*Q][3H] and of Q per head [N*Q][heads][192]
  float *qc32, *qp32;              // ... Q [N*Q][H] and Q' [N][R][E] in f32 (then Q' leaves as (hi | lo) rows [N][R][2E] in qp16)
  unsigned* lncnt;                 // fused residual + LayerNorm: one counter per 64-row tile and problem (zeroed at the start of a forward)
  size_t lncnt_bytes;
  float* s32;                      // scores [N][R][kvp]
  float* stat;                     // split softmax: tile maxima [N][R][ntiles], then tile sums
  float* gfac;                     // split softmax: row factors exp2(m_tile - m_row) / L as [N][ntiles][512] for the P . enc GEMM
  float *st_m, *st_l, *ginv;       // streaming kernels: statistics [N * R][stat_ld], 1 / L [N * R]
  char* gexp;                      // tile factors f16 [N * R][stat_ld]
  int nsplit;
  size_t bytes;
};

// the layer chain's buffers for NN items (residual streams, QKV, attention context, feed-forward intermediate)
void layout_chain(Carver& cv, Work& w, size_t NN, size_t S, size_t H, size_t I, size_t Q) {
  w.hA32 = cv.take<float>(NN * S * H);
  w.hB32 = cv.take<float>(NN * S * H);
  w.pre32 = cv.take<float>(NN * S * H);
  w.hC32 = cv.take<float>(NN * Q * H);
  w.hA16 = cv.take<char>(NN * S * H, 2);
  w.hB16 = cv.take<char>(NN * S * H, 2);
  w.qkv16 = cv.take<char>(NN * S * 3 * H, 2);
  w.ctx16 = cv.take<char>(NN * S * H, 2);
  w.qc16 = cv.take<char>(NN * Q * H, 2);
  w.hC16 = cv.take<char>(NN * Q * H, 2);
  w.ffn16 = cv.take<char>(NN * S * I, 2);
  w.lncnt_bytes = 2 * (NN * S / 64 + 2) * sizeof(unsigned);
  w.lncnt = cv.take<unsigned>(w.lncnt_bytes / sizeof(unsigned));
}

// one handle's cross-attention buffers for N items (K/V cache, or the folded form's Q' / P / U / statistics)
void layout_cross(const mra_qformer* h, Carver& cv, Work& w, int N, int Kv) {
  const mra_cfg& c = h->cfg;
  const size_t H = c.hidden, Q = c.n_query;
  w.kv16 = w.encT = w.qp16 = w.p16 = w.u16 = w.hs16 = w.qs16 = nullptr;
  w.qc32 = w.qp32 = nullptr;
  w.s32 = w.stat = w.gfac = w.st_m = w.st_l = w.ginv = nullptr;
  w.gexp = nullptr;
  if (h->ncross > 0 && use_fold(h, Kv) && fold_streams(h, Kv)) {
    const size_t E = c.enc_width, R = (size_t)c.heads * Q, kvp = fold_kvp(Kv), sld = fold_stream_stat_ld((int)kvp);
    w.qp16 = cv.take<char>((size_t)N * R * E, 2);
    w.encT = cv.take<char>((size_t)N * R * E, 2);     // here: Q' in the streaming kernels' blocked layout
    w.st_m = cv.take<float>((size_t)N * R * sld);
    w.st_l = cv.take<float>((size_t)N * R * sld);
    w.gexp = cv.take<char>((size_t)N * R * sld, 2);
    w.ginv = cv.take<float>((size_t)N * R);
    w.p16 = cv.take<char>((size_t)N * R * kvp, 2);
    w.u16 = cv.take<char>((size_t)N * R * E, 2);
  } else if (h->ncross > 0 && use_fold(h, Kv)) {
    const size_t E = c.enc_width, R = (size_t)c.heads * Q, kvp = fold_kvp(Kv);
    if (!fold_kmajor(h)) w.encT = cv.take<char>((size_t)N * E * kvp, 2);
    w.qp16 = cv.take<char>((size_t)N * R * E * (h->cross_precise ? 2 : 1), 2);
    if (h->cross_precise) {
      w.hs16 = cv.take<char>((size_t)N * Q * 3 * H, 2);
      w.qs16 = cv.take<char>((size_t)N * Q * 3 * H, 2);
      w.qc32 = cv.take<float>((size_t)N * Q * H);
      w.qp32 = cv.take<float>((size_t)N * R * E);
    }
    const bool split = R == 384 && h->sc_tile == 5 && h->split_softmax;   // then the scores never exist in fp32
    if (!split) w.s32 = cv.take<float>((size_t)N * R * kvp);
    w.stat = cv.take<float>((size_t)2 * N * R * ((Kv + 175) / 176));
    if (split && fold_inreg_rescale(h)) w.gfac = cv.take<float>((size_t)N * ((Kv + 175) / 176) * 512);
    w.p16 = cv.take<char>((size_t)N * R * kvp, 2);
    w.u16 = cv.take<char>((size_t)N * R * E, 2);
  } else {
    w.kv16 = cv.take<char>((size_t)h->ncross * 2 * N * Kv * H, 2);
  }
  w.nsplit = attn_pick_split(N, c.heads, (int)Q, Kv);
  w.part = nullptr;
}

Work layout_work(const mra_qformer* h, char* base, int N, int L, int Kv) {
  const mra_cfg& c = h->cfg;
  Carver cv(base);
  Work w;
  layout_chain(cv, w, (size_t)N, (size_t)c.n_query + L, c.hidden, c.inter, c.n_query);
  layout_cross(h, cv, w, N, Kv);   // w.part: grid-split attention partials live right behind `bytes` (set by the caller)
  w.bytes = cv.off;
  return w;
}

// Pair forward: the layer chains of two Q-Formers of equal shape in ONE launch sequence.  The chain buffers hold 2 N items (lane 0's, then
// lane 1's), so attention cores and LayerNorms take both lanes as one launch and every chain GEMM groups the lanes' problems; each lane
// keeps its own cross-attention buffers.
struct PairWork {
  Work w[2];
  long long* mask2;   // attention mask rows of both lanes [2 N][S]
  size_t bytes;
};

PairWork layout_pair(const mra_qformer* h0, const mra_qformer* h1, char* base, int N, int L, int kv0, int kv1) {
  const mra_cfg& c = h0->cfg;
  const size_t H = c.hidden, I = c.inter, Q = c.n_query, S = Q + L;
  Carver cv(base);
  PairWork pw;
  Work sh;
  layout_chain(cv, sh, (size_t)2 * N, S, H, I, Q);
  const mra_qformer* hs[2] = {h0, h1};
  const int kvs[2] = {kv0, kv1};
  for (int l = 0; l < 2; ++l) {
    Work& w = pw.w[l];
    w = sh;
    const size_t n = (size_t)l * N;
    w.hA32 = sh.hA32 ? sh.hA32 + n * S * H : nullptr; w.hB32 = sh.hB32 ? sh.hB32 + n * S * H : nullptr; w.pre32 = sh.pre32 ? sh.pre32 + n * S * H : nullptr;
    w.hC32 = sh.hC32 ? sh.hC32 + n * Q * H : nullptr;
    w.hA16 = sh.hA16 ? sh.hA16 + n * S * H * 2 : nullptr; w.hB16 = sh.hB16 ? sh.hB16 + n * S * H * 2 : nullptr;
    w.qkv16 = sh.qkv16 ? sh.qkv16 + n * S * 3 * H * 2 : nullptr; w.ctx16 = sh.ctx16 ? sh.ctx16 + n * S * H * 2 : nullptr;
    w.qc16 = sh.qc16 ? sh.qc16 + n * Q * H * 2 : nullptr; w.hC16 = sh.hC16 ? sh.hC16 + n * Q * H * 2 : nullptr;
    w.ffn16 = sh.ffn16 ? sh.ffn16 + n * S * I * 2 : nullptr;
    layout_cross(hs[l], cv, w, N, kvs[l]);
    float* part = cv.take<float>(attn_partial_bytes(N, c.heads, (int)Q, w.nsplit) / sizeof(float) + 64);
    w.part = w.nsplit > 1 ? part : nullptr;
    w.bytes = 0;
  }
  pw.mask2 = cv.take<long long>((size_t)2 * N * S);
  pw.bytes = cv.off;
  return pw;
}

// Steps 6 / 6a-6e of one cross-attention layer: from the query projection Q (w.qc16, or w.qc32 in split precision) to the attention
// context (w.ctx16 [N*Q][H]) -- the folded form (per-head Q' GEMM, scores + split softmax, P . enc, per-head context GEMM) or the core over the
// head-major K/V cache.  Shared by mra_qformer_forward and mra_qformer_forward_pair.
int cross_core(mra_qformer* h, const Work& w, const LayerW& Lw, const void* enc, int N, int kv, bool fold, bool stream_fold, bool precise,
               hipStream_t stream) {
  const mra_cfg& c = h->cfg;
  const int Q = c.n_query, H = c.hidden, E = c.enc_width, R = c.heads * Q, kvp = fold_kvp(kv), op = h->op();
  const size_t esz = 2;
  const RowView qc_rows = plain(N * Q, H);
  int rc = 0;
  if (fold) {
    const int ci = Lw.cross_index;
    const bool timed = ci == 0 && h->kv_ev0 && h->kv_ev1;
    if (timed) (void)hipEventRecord(h->kv_ev0, stream);
    // 6a. Q' = Q_h W_k,h per head: [N*32, 64] x [E, 64]^T -> Q' [N][head*32 + q][E]
    GemmProb d{};
    if (precise) {
      // Q (fp32) per head as (hi | lo | hi) over its 64 dims against W_k,h as (hi | hi | lo): K = 192; Q' in fp32, then as (hi | lo) rows
      rc = launch_split_rows(w.qc32, qc_rows, N * Q, H, 64, 3, w.qs16, op, stream);
      if (rc) return chk(rc, "split cross query");
      d.A = w.qs16; d.a = plain(N * Q, 3 * H); d.a_bs = 192;
      d.W = h->arena_p + (size_t)ci * precise_layer_bytes(h) + precise_wk_off(h); d.w_bs = (long long)E * 192;
      d.C = w.qp32; d.c = items_view((long long)R * E, Q, E); d.c_bs_bytes = (long long)Q * E * 4;
      d.M = N * Q; d.N = E; d.K = 192; d.batch = c.heads; d.tile_cfg = (N * Q) % 128 == 0 && E % 128 == 0 ? 2 : 1;
      rc = launch_gemm(&d, 1, EPI_F32, op, stream);
      if (rc) return chk(rc, "fold q' gemm (split precision)");
      rc = launch_split_rows(w.qp32, plain(N * R, E), N * R, E, E, 2, w.qp16, op, stream);
      if (rc) return chk(rc, "split q'");
    } else {
    d.A = w.qc16; d.a = qc_rows; d.a_bs = 64;
    d.W = h->arena_f + (size_t)ci * H * E * esz; d.w_bs = (long long)E * 64;
    d.C = w.qp16; d.c = items_view((long long)R * E, Q, E); d.c_bs_bytes = (long long)Q * E * esz;
    d.M = N * Q; d.N = E; d.K = 64; d.batch = c.heads; d.tile_cfg = (N * Q) % 128 == 0 && E % 128 == 0 ? 2 : 1;
    rc = launch_gemm(&d, 1, EPI_OP, op, stream);
    if (rc) return chk(rc, "fold q' gemm");
    }
    if (stream_fold) {
      // 6b-6d on the streaming kernels (fold_stream.hip): P~ = exp2(s - ceil(tile max)) + tile statistics, row statistics,
      // U = (1 / L) sum g P~ enc with the power-of-two tile factors applied in registers.  P~ is written once, read once.
      FoldStreamArgs fs{};
      fs.qp = w.qp16; fs.qpb = w.encT; fs.enc = enc; fs.p = w.p16; fs.u = w.u16;
      fs.stat_m = w.st_m; fs.stat_l = w.st_l; fs.gexp = w.gexp; fs.ginv = w.ginv;
      fs.items = N; fs.kv = kv; fs.kvp = kvp; fs.E = E;
      fs.alpha = 0.125f * 1.4426950408889634f; fs.phase = 3;
      rc = launch_fold_stream(fs, stream);
      if (rc) return chk(rc, "fold scores / P.enc (streaming kernels)");
    } else {
    // 6b. scores S[n] = Q'[n] enc[n]^T: [R, E] x [kv, E]^T per item, rows padded to kvp columns
    GemmProb sc{};
    sc.A = w.qp16; sc.a = plain(R, E); sc.a_bs = (long long)R * E;
    sc.W = enc; sc.w_bs = (long long)kv * E;
    sc.M = R; sc.N = kv; sc.K = E; sc.batch = N; sc.n_ragged = 1;
    if (precise) {   // Q' rows are (hi | lo): two passes over the same encoder slab inside one K loop
      sc.a = plain(R, 2 * E); sc.a_bs = (long long)R * 2 * E; sc.K = 2 * E; sc.w_kwrap = E / 64;
    }
    sc.tile_cfg = (R == 384 && h->sc_tile == 5) ? 5 : (R == 384 ? h->fold_tile : 2);
    if (sc.tile_cfg == 5 && h->split_softmax) {
      // 6b + 6c fused: the GEMM's epilogue leaves exp2(s - tile maximum) in the operand dtype plus tile statistics;
      // one pass over P rescales every row by exp2(m_tile - m_row) / sum.  The scores never exist in fp32 in HBM.
      const int ntiles = (kv + 175) / 176;
      sc.C = w.p16; sc.c = plain(R, kvp); sc.c_bs_bytes = (long long)R * kvp * esz;
      sc.alpha = 0.125f * 1.4426950408889634f;
      sc.stat_m = w.stat; sc.stat_l = w.stat + (size_t)N * R * ntiles;
      rc = launch_gemm(&sc, 1, EPI_SOFTPART, op, stream);
      if (rc) return chk(rc, "fold scores gemm (softmax partials)");
      if (w.gfac) {
        // second half of the softmax without a pass over P: only the row factors are computed here, the P . enc GEMM applies them
        // to its P~ fragments in registers (same arithmetic, same rounding as the rescale pass)
        rc = launch_fold_rowfactor(sc.stat_m, sc.stat_l, w.gfac, N * R, R, ntiles, w.p16, kvp, 176, kvp, stream);
        if (rc) return chk(rc, "fold row factors");
      } else {
        rc = launch_softmax_rescale(w.p16, kvp, sc.stat_m, sc.stat_l, N * R, ntiles, 176, kvp, op, stream);
        if (rc) return chk(rc, "fold softmax rescale");
      }
    } else {
      sc.C = w.s32; sc.c = plain(R, kvp); sc.c_bs_bytes = (long long)R * kvp * 4;   // fp32 rows
      rc = launch_gemm(&sc, 1, EPI_F32, op, stream);
      if (rc) return chk(rc, "fold scores gemm");
      // 6c. P = softmax(S / 8) row by row (the key bias is constant along a row and cancels)
      rc = launch_softmax_rows(w.s32, kvp, w.p16, kvp, N * R, kv, kvp, 0.125f, op, stream);
      if (rc) return chk(rc, "fold softmax");
    }
    // 6d. U[n] = P[n] enc[n]: [R, kvp] x [E, kvp]^T per item
    GemmProb pv{};
    pv.A = w.p16; pv.a = plain(R, kvp); pv.a_bs = (long long)R * kvp;
    if (fold_kmajor(h)) {   // the encoder tokens themselves: [kv][E] is W K-major; rows kv .. kvp repeat the last token against P = 0
      pv.W = enc; pv.w_bs = (long long)kv * E; pv.w_ld = E; pv.k_rows = kv;
    } else {
      pv.W = w.encT; pv.w_bs = (long long)E * kvp;
    }
    pv.C = w.u16; pv.c = plain(R, E); pv.c_bs_bytes = (long long)R * E * esz;
    pv.M = R; pv.N = E; pv.K = kvp; pv.batch = N;
    pv.tile_cfg = (R == 384 && E % 176 == 0 && h->pv_tile == 5) ? 5 : (R == 384 ? h->fold_tile : 2);
    if (w.gfac && sc.tile_cfg == 5 && h->split_softmax) { pv.pscale = w.gfac; pv.ps_ntiles = (kv + 175) / 176; }
    rc = launch_gemm(&pv, 1, EPI_OP, op, stream);
    if (rc) return chk(rc, "fold p.enc gemm");
    }
    // 6e. context = U_h W_v,h^T + b_v,h per head: [N*32, E] x [64, E]^T -> ctx [N*32][head*64 + d]
    GemmProb cx{};
    cx.A = w.u16; cx.a = items_view((long long)R * E, Q, E); cx.a_bs = (long long)Q * E;
    cx.W = (const char*)h->wkv + (size_t)(ci * 2 + 1) * H * E * esz; cx.w_bs = (long long)64 * E;
    cx.bias = h->bkv + (size_t)(ci * 2 + 1) * H; cx.bias_bs = 64;
    cx.C = w.ctx16; cx.c = qc_rows; cx.c_bs_bytes = 64 * esz;
    cx.M = N * Q; cx.N = 64; cx.K = E; cx.batch = c.heads; cx.tile_cfg = E % 128 == 0 && N * Q >= 512 ? 6 : 1;
    rc = launch_gemm(&cx, 1, EPI_OP, op, stream);
    if (rc) return chk(rc, "fold context gemm");
    if (timed) (void)hipEventRecord(h->kv_ev1, stream);
  } else {
  // 6. cross-attention core over the head-major K/V cache
  AttnArgs a{};
  const size_t per_sel = (size_t)N * c.heads * kv * 64;
  a.Q = w.qc16;
  a.K = w.kv16 + (size_t)(Lw.cross_index * 2 + 0) * per_sel * esz;
  a.V = w.kv16 + (size_t)(Lw.cross_index * 2 + 1) * per_sel * esz;
  a.O = w.ctx16;
  a.q_item_stride = (long long)Q * H; a.q_ld = H;
  a.k_item_stride = (long long)c.heads * kv * 64; a.k_head_stride = (long long)kv * 64; a.k_ld = 64;
  a.v_item_stride = a.k_item_stride; a.v_head_stride = a.k_head_stride; a.v_ld = 64;
  a.o_item_stride = (long long)Q * H; a.o_ld = H;
  a.mask = nullptr; a.mask_ld = 0;
  a.items = N; a.heads = c.heads; a.q_rows = Q; a.kv_len = kv;
  a.scale = 0.125f; a.nsplit = w.nsplit; a.part = w.part;
  rc = launch_attention(a, op, stream);
  if (rc) return chk(rc, "cross attention");
  }
  return MRA_OK;
}


}  // namespace

namespace mra_host {
// K/V of every cross layer in ONE GEMM: [items*kv, E] x [ncross*2*H, E]^T, scattered head-major.
int kv_project(const mra_qformer* h, const void* enc, int N, int kv, void* kv_cache, hipStream_t stream) {
  const mra_cfg& c = h->cfg;
  GemmProb p{};
  p.A = enc;
  p.a = plain(N * kv, c.enc_width);
  p.W = h->wkv;
  p.bias = h->bkv;
  p.C = kv_cache;
  p.c = plain(1, 1);
  p.M = N * kv;
  p.N = h->ncross * 2 * c.hidden;
  p.K = c.enc_width;
  p.kv_tokens = kv;
  p.kv_items = N;
  p.kv_heads = c.heads;
  p.persist = 1;   // on the eight-phase kernel: one persistent workgroup per CU (bit-identical; no workgroup turnaround between tiles)
  return launch_gemm(&p, 1, EPI_KV, h->op(), stream);
}

}  // namespace mra_host

extern "C" {

const char* mra_last_error(void) { return g_err.c_str(); }
const char* mra_version(void) { return "mraudio_amd 0.1 (gfx950)"; }
int64_t mra_debug_gemm_launches(int32_t family, int32_t epilogue) { return gemm_launch_count(family, epilogue); }

void mra_cfg_default(mra_cfg* c, int32_t enc_width) {
  c->hidden = 768;
  c->heads = 12;
  c->inter = 3072;
  c->layers = 12;
  c->cross_freq = 2;
  c->enc_width = enc_width;
  c->n_query = 32;
  c->vocab = 30523;
  c->max_pos = 512;
  c->ln_eps = 1e-12f;
  c->enc_ln_eps = 1e-5f;
  c->llm_hidden = 4096;
  c->op_dtype = MRA_F16;
}

int mra_qformer_create(const mra_cfg* cfg, mra_qformer** out) {
  if (!cfg || !out) return fail(MRA_EINVAL, "null argument");
  const mra_cfg& c = *cfg;
  if (c.hidden <= 0 || c.hidden % 256 || c.hidden > 1024) return fail(MRA_EINVAL, "hidden must be a multiple of 256, <= 1024");
  if (c.heads <= 0 || c.hidden != c.heads * 64) return fail(MRA_EINVAL, "head_dim must be 64");
  if (c.n_query != 32) return fail(MRA_EINVAL, "n_query must be 32");
  if (c.inter <= 0 || c.inter % 256) return fail(MRA_EINVAL, "inter must be a multiple of 256");
  if (c.enc_width <= 0 || c.enc_width % 64) return fail(MRA_EINVAL, "enc_width must be a multiple of 64");
  if (c.layers <= 0 || c.cross_freq <= 0 || c.vocab <= 0 || c.max_pos <= 0) return fail(MRA_EINVAL, "bad layer/vocab config");
  if (c.llm_hidden < 0 || c.llm_hidden % 256) return fail(MRA_EINVAL, "llm_hidden must be 0 or a multiple of 256");
  if (c.op_dtype != MRA_F16 && c.op_dtype != MRA_BF16) return fail(MRA_EINVAL, "op_dtype must be MRA_F16 or MRA_BF16");
  mra_qformer* h = new mra_qformer();
  h->cfg = c;
  HIP_TRY(hipGetDevice(&h->device));
  h->arena_bytes = layout_params(h, nullptr);
  hipError_t e = hipMalloc((void**)&h->arena, h->arena_bytes);
  if (e != hipSuccess) {
    delete h;
    return fail(MRA_ENOMEM, std::string("hipMalloc of parameter arena: ") + hipGetErrorString(e));
  }
  layout_params(h, h->arena);
  if (h->ncross > 0) {
    e = hipMalloc((void**)&h->arena_f, (size_t)h->ncross * c.hidden * c.enc_width * 2);
    if (e != hipSuccess) {
      mra_qformer_destroy(h);
      return fail(MRA_ENOMEM, std::string("fold weight arena: ") + hipGetErrorString(e));
    }
  }
  // segment table of mra_qformer_load_flat: every bert.* parameter in chunks of FLAT_SEG elements
  std::vector<FlatSeg> segs;
  for (auto& kv : h->params) {
    if (kv.first.rfind("bert.", 0) != 0) continue;
    const Param& pr = kv.second;
    const size_t esz = pr.dtype == MRA_F32 ? 4 : 2;
    for (long long o = 0; o < pr.numel; o += FLAT_SEG)
      segs.push_back(FlatSeg{(unsigned long long)(pr.goff / 4 + o), (char*)pr.ptr + o * esz,
                             (int)std::min<long long>(FLAT_SEG, pr.numel - o), pr.dtype});
    if (pr.copy32)   // the f32 copies of the score-chain weights follow the master too
      for (long long o = 0; o < pr.numel; o += FLAT_SEG)
        segs.push_back(FlatSeg{(unsigned long long)(pr.goff / 4 + o), (char*)(pr.copy32 + o), (int)std::min<long long>(FLAT_SEG, pr.numel - o), MRA_F32});
  }
  h->n_flat_segs = (int)segs.size();
  e = hipMalloc((void**)&h->flat_segs, segs.size() * sizeof(FlatSeg));
  if (e == hipSuccess) e = hipMemcpy(h->flat_segs, segs.data(), segs.size() * sizeof(FlatSeg), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    mra_qformer_destroy(h);
    return fail(MRA_ENOMEM, std::string("segment table: ") + hipGetErrorString(e));
  }
  *out = h;
  return MRA_OK;
}

void mra_qformer_destroy(mra_qformer* h) {
  if (!h) return;
  if (h->arena) (void)hipFree(h->arena);
  if (h->arena_t) (void)hipFree(h->arena_t);
  if (h->arena_f) (void)hipFree(h->arena_f);
  if (h->arena_p) (void)hipFree(h->arena_p);
  if (h->flat_segs) (void)hipFree(h->flat_segs);
  if (h->tr_jobs) (void)hipFree(h->tr_jobs);
  if (h->adam_jobs) (void)hipFree(h->adam_jobs);
  if (h->adam_segs) (void)hipFree(h->adam_segs);
  if (h->c32_segs) (void)hipFree(h->c32_segs);
  for (auto& e : h->wg_ev) if (e) (void)hipEventDestroy(e);
  if (h->wg_stream) (void)hipStreamDestroy(h->wg_stream);
  delete h;
}

int mra_qformer_load(mra_qformer* h, const char* name, const void* src, int32_t dtype, const int64_t* shape,
                     int32_t ndim, void* stream) {
  if (!h || !name || !src || (ndim > 0 && !shape)) return fail(MRA_EINVAL, "null argument");
  if (dtype < MRA_F32 || dtype > MRA_BF16) return fail(MRA_EINVAL, "bad dtype");
  const std::string key(name);
  if (key == "bert.embeddings.position_ids") return MRA_OK;  // LAVIS buffer, not a parameter
  auto it = h->params.find(key);
  if (it == h->params.end()) return fail(MRA_ENAME, "unknown parameter name: " + key);
  long long numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= shape[i];
  if (numel != it->second.numel)
    return fail(MRA_EINVAL, "parameter " + key + ": expected " + std::to_string(it->second.numel) + " elements, got " +
                                std::to_string(numel));
  int rc = launch_convert(src, dtype, it->second.ptr, it->second.dtype, numel, as_stream(stream));
  if (!rc && it->second.copy32) rc = launch_convert(src, dtype, it->second.copy32, MRA_F32, numel, as_stream(stream));
  if (rc) return chk(rc, "launch_convert");
  it->second.loaded = true;
  h->transposes_stale = true;
  h->fold_stale = true;
  h->precise_stale = true;
  return MRA_OK;
}

int mra_qformer_load_flat(mra_qformer* h, const float* master, size_t master_bytes, void* stream) {
  if (!h || !master) return fail(MRA_EINVAL, "null argument");
  if (master_bytes < h->grad_bytes) return fail(MRA_EINVAL, "master buffer smaller than mra_qformer_grad_bytes()");
  if ((size_t)master & 15) return fail(MRA_EINVAL, "master buffer must be 16-byte aligned");
  const int rc = launch_convert_flat(master, h->flat_segs, h->n_flat_segs, as_stream(stream));
  if (rc) return chk(rc, "launch_convert_flat");
  for (auto& kv : h->params)
    if (kv.first.rfind("bert.", 0) == 0) kv.second.loaded = true;
  h->transposes_stale = true;
  h->fold_stale = true;
  h->precise_stale = true;
  return MRA_OK;
}

int mra_qformer_missing(mra_qformer* h, char* buf, size_t buflen) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  int n = 0;
  std::string s;
  for (auto& kv : h->params)
    if (!kv.second.loaded) {
      ++n;
      if (!s.empty()) s += ",";
      s += kv.first;
    }
  if (buf && buflen) {
    std::strncpy(buf, s.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
  }
  return n;
}

int mra_modality_ln(mra_qformer* h, const void* x, int32_t x_dtype, const int64_t* item_index, int32_t items,
                    int32_t tokens, void* out, void* stream) {
  if (!h || (!x && items > 0) || (!out && items > 0)) return fail(MRA_EINVAL, "null argument");
  if (items < 0 || tokens < 0) return fail(MRA_EINVAL, "negative size");
  if (items == 0 || tokens == 0) return MRA_OK;
  if (!h->params["ln.weight"].loaded || !h->params["ln.bias"].loaded) return fail(MRA_ESTATE, "ln.weight / ln.bias not loaded");
  return chk(launch_modality_ln(x, x_dtype, (const long long*)item_index, items, tokens, h->cfg.enc_width, h->encg,
                                h->encb, h->cfg.enc_ln_eps, out, h->op(), as_stream(stream)),
             "modality_ln");
}

size_t mra_qformer_workspace_bytes(mra_qformer* h, int32_t items, int32_t L, int32_t kv) {
  if (!h || items <= 0 || L < 0 || kv <= 0) return 0;
  Work w = layout_work(h, nullptr, items, L, kv);
  return w.bytes + align_up(attn_partial_bytes(items, h->cfg.heads, h->cfg.n_query, w.nsplit));
}

double mra_qformer_flops(mra_qformer* h, int32_t items, int32_t L, int32_t kv, int32_t with_last_text) {
  if (!h) return 0.0;
  const mra_cfg& c = h->cfg;
  const double H = c.hidden, I = c.inter, E = c.enc_width, Q = c.n_query, S = Q + L, Kv = kv;
  double per_item = c.layers * (8 * S * H * H + 4 * S * S * H + 4 * S * H * I) +
                    h->ncross * (4 * Q * H * H + 4 * Kv * E * H + 4 * Q * Kv * H);
  if (!with_last_text) per_item -= 4.0 * L * H * I;
  return per_item * items;
}

int mra_qformer_forward(mra_qformer* h, const int64_t* input_ids, const int64_t* attention_mask,
                        const float* query_embeds, int32_t query_items, const void* enc, int32_t items, int32_t L,
                        int32_t kv, float* out_query, float* out_full, float* out_cls, void* workspace,
                        size_t workspace_bytes, void* stream_) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (items < 0 || L < 0 || kv < 0) return fail(MRA_EINVAL, "negative size");
  if (items == 0) return MRA_OK;
  const mra_cfg& c = h->cfg;
  if (kv == 0) return fail(MRA_EINVAL, "kv must be >= 1");
  if (L > c.max_pos) return fail(MRA_EINVAL, "L exceeds max_pos");
  if (!enc || (L > 0 && !input_ids)) return fail(MRA_EINVAL, "null input");
  if (out_cls && L < 1) return fail(MRA_EINVAL, "out_cls needs L >= 1");
  if (query_embeds && query_items != 1 && query_items != items)
    return fail(MRA_EINVAL, "query_items must be 1 or items");
  if (!out_query && !out_full && !out_cls) return fail(MRA_EINVAL, "no output requested");
  {
    char names[256];
    const int miss = mra_qformer_missing(h, names, sizeof(names));
    int tolerated = 0;
    for (const char* opt : {"ln.weight", "ln.bias", "llm_proj.weight", "llm_proj.bias"}) {
      auto it = h->params.find(opt);
      if (it != h->params.end() && !it->second.loaded) ++tolerated;
    }
    if (miss > tolerated) return fail(MRA_ESTATE, std::string("parameters not loaded: ") + names);
  }
  const size_t need = mra_qformer_workspace_bytes(h, items, L, kv);
  if (!workspace || workspace_bytes < need)
    return fail(MRA_ENOMEM, "workspace too small: need " + std::to_string(need) + " bytes");
  if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MRA_EINVAL, "workspace must be 256-byte aligned");

  hipStream_t stream = as_stream(stream_);
  const int N = items, Q = c.n_query, S = Q + L, H = c.hidden, I = c.inter, E = c.enc_width;
  const int op = h->op();
  Work w = layout_work(h, (char*)workspace, N, L, kv);
  w.part = w.nsplit > 1 ? reinterpret_cast<float*>((char*)workspace + w.bytes) : nullptr;
  const long long SH = (long long)S * H;
  const size_t esz = 2;

  // row views of the [N, S, H] streams
  const RowView all_rows = plain(N * S, H);
  const RowView q_view = items_view(SH, Q, H);          // rows [:, :32]
  const RowView t_view = items_view(SH, L > 0 ? L : 1, H);  // rows [:, 32:] (base pointer + 32*H)
  const RowView cls_view = items_view(SH, 1, H);        // row  [:, 32]
  const RowView qc_rows = plain(N * Q, H);              // compact [N*32, H]

  // embeddings
  const float* qsrc = query_embeds ? query_embeds : h->query;
  const long long qstride = query_embeds && query_items == items && items > 1 ? (long long)Q * H : 0;
  int rc = launch_embed_ln((const long long*)input_ids, N, L, Q, H, c.vocab, qsrc, qstride, h->word, h->pos, h->embg,
                           h->embb, c.ln_eps, w.hA32, w.hA16, nullptr, op, stream);
  if (rc) return chk(rc, "embed_ln");

  // residual projection + LayerNorm in ONE launch (chain_ring bit 3, on the 96 x 64 ring tile): its per-row-tile counters start at zero
  const bool ln_fuse = (h->chain_ring & 12) == 12 && H % 96 == 0 && H % 256 == 0 && H <= 1024;
  if (ln_fuse) HIP_TRY(hipMemsetAsync(w.lncnt, 0, w.lncnt_bytes, stream));
  const bool fold = h->ncross > 0 && use_fold(h, kv);
  const bool stream_fold = fold && fold_streams(h, kv);
  const int R = c.heads * Q, kvp = fold_kvp(kv);
  if (fold) {
    // folded cross-attention: enc^T per item (the K-contiguous operand of P . enc), key weights regrouped per head
    if (!stream_fold && !fold_kmajor(h)) {
      rc = launch_transpose_pad(enc, w.encT, kv, E, kvp, (long long)kv * E, (long long)E * kvp, N, op, stream);
      if (rc) return chk(rc, "enc transpose");
    }
    if ((rc = mra_qformer_prepare(h, stream_))) return rc;
  } else if (h->ncross > 0) {
    // K/V of every cross layer in one GEMM, scattered head-major
    if (h->kv_ev0 && h->kv_ev1) (void)hipEventRecord(h->kv_ev0, stream);
    rc = kv_project(h, enc, N, kv, w.kv16, stream);
    if (rc) return chk(rc, "kv projection gemm");
    if (h->kv_ev0 && h->kv_ev1) (void)hipEventRecord(h->kv_ev1, stream);
  }
  if (h->kv_done) (void)hipEventRecord(h->kv_done, stream);   // also without cross layers: the waiter must not hang

  const bool want_text_last = out_full != nullptr;
  const bool want_cls_last = !out_full && out_cls != nullptr;

  for (int i = 0; i < c.layers; ++i) {
    const LayerW& Lw = h->layers[i];
    const bool last = i == c.layers - 1;
    // 1. fused Q|K|V projection of all S rows
    {
      GemmProb p{};
      p.A = w.hA16; p.a = all_rows;
      p.W = Lw.wqkv; p.bias = Lw.bqkv;
      p.C = w.qkv16; p.c = plain(N * S, 3 * H);
      p.M = N * S; p.N = 3 * H; p.K = H;
      // The chain GEMMs at ~1-2 k rows are bound by the operand bytes each CU pulls through its load path (~33 B / clk from L2),
      // not by tile count: 128 x 128 tiles (288 of them) move half the bytes per flop of the 1152 64 x 64 tiles the automatic
      // choice makes (headline step 6.75 -> 6.67 ms, reference item shape 2.65 -> 2.55 ms together with the down-projection below)
      p.tile_cfg = N * S >= 1024 ? 2 : 0;
      // ... and at ~2 k rows the ring kernel's 144 x 128 tile is exactly one workgroup per CU (2048 x 2304 = 16 x 16 tiles): 14.4 vs 17.6 us stand-alone
      if ((h->chain_ring & 1) && N * S >= 1024 && (3 * H) % 144 == 0) p.tile_cfg = 9;
      rc = launch_gemm(&p, 1, EPI_OP, op, stream);
      if (rc) return chk(rc, "qkv gemm");
    }
    // 2. self-attention core
    {
      AttnArgs a{};
      a.Q = w.qkv16;
      a.K = w.qkv16 + (size_t)H * esz;
      a.V = w.qkv16 + (size_t)2 * H * esz;
      a.O = w.ctx16;
      a.q_item_stride = (long long)S * 3 * H; a.q_ld = 3 * H;
      a.k_item_stride = (long long)S * 3 * H; a.k_head_stride = 64; a.k_ld = 3 * H;
      a.v_item_stride = (long long)S * 3 * H; a.v_head_stride = 64; a.v_ld = 3 * H;
      a.o_item_stride = SH; a.o_ld = H;
      a.mask = (const long long*)attention_mask; a.mask_ld = S;
      a.items = N; a.heads = c.heads; a.q_rows = S; a.kv_len = S;
      a.scale = 0.125f; a.nsplit = 1; a.part = nullptr;
      rc = launch_attention(a, op, stream);
      if (rc) return chk(rc, "self attention");
    }
    // 3. output projection + residual, 4. LayerNorm -> hB
    {
      GemmProb p{};
      p.A = w.ctx16; p.a = all_rows;
      p.W = Lw.wo; p.bias = Lw.bo;
      p.R = w.hA32; p.r = all_rows;
      p.C = w.pre32; p.c = all_rows;
      p.M = N * S; p.N = H; p.K = H;
      if ((h->chain_ring & 4) && N * S >= 1024 && H % 96 == 0) p.tile_cfg = 11;
      if (ln_fuse && p.tile_cfg == 11) {
        // HF:519-530 in one launch: the column tile of a 64-row block that finishes last normalises the block's rows
        p.ln_gain = Lw.ln1g; p.ln_bias = Lw.ln1b; p.ln_eps = c.ln_eps;
        p.ln_y32 = w.hB32; p.ln_y32v = all_rows; p.ln_y16 = w.hB16; p.ln_y16v = all_rows; p.ln_counter = w.lncnt;
        rc = launch_gemm(&p, 1, EPI_RES_LN, op, stream);
        if (rc) return chk(rc, "attn out gemm + ln");
      } else {
      rc = launch_gemm(&p, 1, EPI_RES_F32, op, stream);
      if (rc) return chk(rc, "attn out gemm");
      rc = launch_ln_rows(w.pre32, all_rows, N * S, H, Lw.ln1g, Lw.ln1b, c.ln_eps, w.hB32, all_rows, w.hB16, all_rows, op,
                          stream);
      if (rc) return chk(rc, "attn ln");
      }
    }
    // query-side state entering the feed-forward: hB[:, :32] or the cross-attention output hC
    const void* fq16 = w.hB16;
    const float* fq32 = w.hB32;
    RowView fqv = q_view;
    if (Lw.cross_index >= 0) {
      const bool precise = fold && h->cross_precise;
      // 5. cross query projection
      GemmProb p{};
      if (precise) {
        // split precision: the fp32 query rows leave as (hi | lo | hi), the weight is stored as (hi | hi | lo): one GEMM over K = 3H
        // computes xh wh + xl wh + xh wl in the fp32 accumulators; Q stays fp32
        rc = launch_split_rows(w.hB32, q_view, N * Q, H, H, 3, w.hs16, op, stream);
        if (rc) return chk(rc, "split query rows");
        p.A = w.hs16; p.a = plain(N * Q, 3 * H);
        p.W = h->arena_p + (size_t)Lw.cross_index * precise_layer_bytes(h); p.bias = Lw.bcq;
        p.C = w.qc32; p.c = qc_rows;
        p.M = N * Q; p.N = H; p.K = 3 * H;
        rc = launch_gemm(&p, 1, EPI_F32, op, stream);
      } else {
        p.A = w.hB16; p.a = q_view;
        p.W = Lw.wcq; p.bias = Lw.bcq;
        p.C = w.qc16; p.c = qc_rows;
        p.M = N * Q; p.N = H; p.K = H;
        rc = launch_gemm(&p, 1, EPI_OP, op, stream);
      }
      if (rc) return chk(rc, "cross q gemm");
      if ((rc = cross_core(h, w, Lw, enc, N, kv, fold, stream_fold, precise, stream))) return rc;
      // 7. output projection + residual (hB[:, :32]), 8. LayerNorm -> hC (compact)
      GemmProb o{};
      o.A = w.ctx16; o.a = qc_rows;
      o.W = Lw.wco; o.bias = Lw.bco;
      o.R = w.hB32; o.r = q_view;
      o.C = w.pre32; o.c = qc_rows;
      o.M = N * Q; o.N = H; o.K = H;
      if (ln_fuse && N * Q >= 512) {
        o.tile_cfg = 11;
        o.ln_gain = Lw.lncg; o.ln_bias = Lw.lncb; o.ln_eps = c.ln_eps;
        o.ln_y32 = w.hC32; o.ln_y32v = qc_rows; o.ln_y16 = w.hC16; o.ln_y16v = qc_rows; o.ln_counter = w.lncnt;
        rc = launch_gemm(&o, 1, EPI_RES_LN, op, stream);
        if (rc) return chk(rc, "cross out gemm + ln");
      } else {
      rc = launch_gemm(&o, 1, EPI_RES_F32, op, stream);
      if (rc) return chk(rc, "cross out gemm");
      rc = launch_ln_rows(w.pre32, qc_rows, N * Q, H, Lw.lncg, Lw.lncb, c.ln_eps, w.hC32, qc_rows, w.hC16, qc_rows, op,
                          stream);
      if (rc) return chk(rc, "cross ln");
      }
      fq16 = w.hC16;
      fq32 = w.hC32;
      fqv = qc_rows;
    }
    // 9-14. feed-forwards: group 0 = query rows, group 1 = text rows (own weights)
    int text_rows = 0;
    RowView tv = t_view;
    if (L > 0) {
      if (!last || want_text_last) text_rows = N * L;
      else if (want_cls_last) { text_rows = N; tv = cls_view; }
    }
    bool ffn_ln_fused = false;
    const size_t t_off16 = (size_t)Q * H * esz;   // byte offset of row 32 inside an item (op dtype)
    const size_t t_off32 = (size_t)Q * H;         // element offset (f32)
    {
      GemmProb g[2] = {};
      g[0].A = fq16; g[0].a = fqv;
      g[0].W = Lw.wiq; g[0].bias = Lw.biq;
      g[0].C = w.ffn16; g[0].c = plain(N * Q, I);
      g[0].M = N * Q; g[0].N = I; g[0].K = H;
      g[1].A = w.hB16 + t_off16; g[1].a = tv;
      g[1].W = Lw.wit; g[1].bias = Lw.bit;
      g[1].C = w.ffn16 + (size_t)N * Q * I * esz; g[1].c = plain(text_rows, I);
      g[1].M = text_rows; g[1].N = I; g[1].K = H;
      if ((h->chain_ring & 2) && N * Q >= 512 && I % 192 == 0) g[0].tile_cfg = 10;
      rc = launch_gemm(g, text_rows > 0 ? 2 : 1, EPI_GELU_OP, op, stream);
      if (rc) return chk(rc, "ffn up gemm");
    }
    {
      GemmProb g[2] = {};
      g[0].A = w.ffn16; g[0].a = plain(N * Q, I);
      g[0].W = Lw.woq; g[0].bias = Lw.boq;
      g[0].R = fq32; g[0].r = fqv;
      g[0].C = w.pre32; g[0].c = q_view;
      g[0].M = N * Q; g[0].N = H; g[0].K = I;
      g[1].A = w.ffn16 + (size_t)N * Q * I * esz; g[1].a = plain(text_rows, I);
      g[1].W = Lw.wot; g[1].bias = Lw.bot;
      g[1].R = w.hB32 + t_off32; g[1].r = tv;
      g[1].C = w.pre32 + t_off32; g[1].c = tv;
      g[1].M = text_rows; g[1].N = H; g[1].K = I;
      g[0].tile_cfg = N * Q >= 512 && I % 128 == 0 ? 6 : 0;   // 64 weight rows x 128 activation rows, 128-deep K steps (see the QKV note)
      if ((h->chain_ring & 4) && N * Q >= 512 && H % 96 == 0) g[0].tile_cfg = 11;
      ffn_ln_fused = ln_fuse && !last && g[0].tile_cfg == 11;
      if (ffn_ln_fused) {
        // HF:573-587 for both row sets in the same launch: problem 0 = query rows (output_query.LayerNorm), problem 1 = text rows (output.LayerNorm)
        g[0].ln_gain = Lw.lnqg; g[0].ln_bias = Lw.lnqb; g[0].ln_eps = c.ln_eps;
        g[0].ln_y32 = w.hA32; g[0].ln_y32v = q_view; g[0].ln_y16 = w.hA16; g[0].ln_y16v = q_view; g[0].ln_counter = w.lncnt;
        g[1].ln_gain = Lw.lntg; g[1].ln_bias = Lw.lntb; g[1].ln_eps = c.ln_eps;
        g[1].ln_y32 = w.hA32 + t_off32; g[1].ln_y32v = tv; g[1].ln_y16 = w.hA16 + t_off16; g[1].ln_y16v = tv;
        g[1].ln_counter = w.lncnt + (N * Q + 63) / 64;
        rc = launch_gemm(g, text_rows > 0 ? 2 : 1, EPI_RES_LN, op, stream);
        if (rc) return chk(rc, "ffn down gemm + ln");
      } else {
      rc = launch_gemm(g, text_rows > 0 ? 2 : 1, EPI_RES_F32, op, stream);
      if (rc) return chk(rc, "ffn down gemm");
      }
    }
    if (ffn_ln_fused) {
      // the LayerNorms ran inside the down-projection's launch
    } else if (!last && text_rows == N * L && L > 0) {
      // query and text LayerNorm in one launch: the two row sets are the whole [N, S, H] stream
      rc = launch_ln_rows2(w.pre32, all_rows, N * S, H, Lw.lnqg, Lw.lnqb, Lw.lntg, Lw.lntb, S, Q, c.ln_eps, w.hA32, all_rows,
                           w.hA16, all_rows, op, stream);
      if (rc) return chk(rc, "ffn ln");
    } else if (!last) {
      rc = launch_ln_rows(w.pre32, q_view, N * Q, H, Lw.lnqg, Lw.lnqb, c.ln_eps, w.hA32, q_view, w.hA16, q_view, op, stream);
      if (rc) return chk(rc, "ffn query ln");
      if (text_rows > 0) {
        rc = launch_ln_rows(w.pre32 + t_off32, tv, text_rows, H, Lw.lntg, Lw.lntb, c.ln_eps, w.hA32 + t_off32, tv,
                            w.hA16 + t_off16, tv, op, stream);
        if (rc) return chk(rc, "ffn text ln");
      }
    } else {
      // last layer: LayerNorm straight into the caller's buffers
      if (out_full) {
        rc = launch_ln_rows(w.pre32, q_view, N * Q, H, Lw.lnqg, Lw.lnqb, c.ln_eps, out_full, q_view, nullptr, q_view, op, stream);
        if (rc) return chk(rc, "final query ln");
        if (text_rows > 0) {
          rc = launch_ln_rows(w.pre32 + t_off32, tv, text_rows, H, Lw.lntg, Lw.lntb, c.ln_eps, out_full + t_off32, tv, nullptr,
                              tv, op, stream);
          if (rc) return chk(rc, "final text ln");
        }
        if (out_query) {
          rc = launch_copy_rows_f32(out_full, q_view, out_query, qc_rows, N * Q, H, stream);
          if (rc) return chk(rc, "copy out_query");
        }
        if (out_cls) {
          rc = launch_copy_rows_f32(out_full + t_off32, cls_view, out_cls, plain(N, H), N, H, stream);
          if (rc) return chk(rc, "copy out_cls");
        }
      } else {
        if (out_query) {
          rc = launch_ln_rows(w.pre32, q_view, N * Q, H, Lw.lnqg, Lw.lnqb, c.ln_eps, out_query, qc_rows, nullptr, qc_rows, op,
                              stream);
          if (rc) return chk(rc, "final query ln");
        }
        if (out_cls) {
          rc = launch_ln_rows(w.pre32 + t_off32, cls_view, N, H, Lw.lntg, Lw.lntb, c.ln_eps, out_cls, plain(N, H), nullptr,
                              plain(N, H), op, stream);
          if (rc) return chk(rc, "final cls ln");
        }
      }
    }
  }
  return MRA_OK;
}

size_t mra_qformer_pair_workspace_bytes(mra_qformer* h0, mra_qformer* h1, int32_t items, int32_t L, int32_t kv0, int32_t kv1) {
  if (!h0 || !h1 || items <= 0 || L < 0 || kv0 <= 0 || kv1 <= 0) return 0;
  return layout_pair(h0, h1, nullptr, items, L, kv0, kv1).bytes;
}

int mra_qformer_forward_pair(mra_qformer* h0, mra_qformer* h1, const int64_t* input_ids, const int64_t* attention_mask, const void* enc0,
                             const void* enc1, int32_t items, int32_t L, int32_t kv0, int32_t kv1, float* out_query0, float* out_cls0,
                             float* out_query1, float* out_cls1, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!h0 || !h1) return fail(MRA_EINVAL, "null handle");
  if (h0 == h1) return fail(MRA_EINVAL, "the two lanes need two handles");
  if (items < 0 || L < 0 || kv0 < 0 || kv1 < 0) return fail(MRA_EINVAL, "negative size");
  if (items == 0) return MRA_OK;
  mra_qformer* hs[2] = {h0, h1};
  const void* encs[2] = {enc0, enc1};
  const int kvs[2] = {kv0, kv1};
  float* outq[2] = {out_query0, out_query1};
  float* outc[2] = {out_cls0, out_cls1};
  const mra_cfg& c = h0->cfg;
  {
    const mra_cfg& d = h1->cfg;
    if (c.hidden != d.hidden || c.heads != d.heads || c.inter != d.inter || c.layers != d.layers || c.cross_freq != d.cross_freq || c.n_query != d.n_query ||
        c.op_dtype != d.op_dtype || c.ln_eps != d.ln_eps || h0->device != h1->device)
      return fail(MRA_EINVAL, "pair forward: the two Q-Formers must agree in hidden / heads / inter / layers / cross_freq / n_query / op_dtype and live on one device");
  }
  if (kv0 == 0 || kv1 == 0) return fail(MRA_EINVAL, "kv must be >= 1");
  if (L > c.max_pos || L > h1->cfg.max_pos) return fail(MRA_EINVAL, "L exceeds max_pos");
  if (!enc0 || !enc1 || (L > 0 && !input_ids)) return fail(MRA_EINVAL, "null input");
  if ((out_cls0 || out_cls1) && L < 1) return fail(MRA_EINVAL, "out_cls needs L >= 1");
  for (int l = 0; l < 2; ++l) {
    if (!outq[l] && !outc[l]) return fail(MRA_EINVAL, "no output requested for a lane");
    if (hs[l]->cross_precise) return fail(MRA_ESTATE, "pair forward runs the operand-dtype score chain: use mra_qformer_forward for split precision");
    if (hs[l]->ncross > 0 && use_fold(hs[l], kvs[l]) && fold_streams(hs[l], kvs[l])) return fail(MRA_ESTATE, "pair forward: the streaming fold kernels are not supported");
    char names[256];
    const int miss = mra_qformer_missing(hs[l], names, sizeof(names));
    int tolerated = 0;
    for (const char* opt : {"ln.weight", "ln.bias", "llm_proj.weight", "llm_proj.bias"}) {
      auto it = hs[l]->params.find(opt);
      if (it != hs[l]->params.end() && !it->second.loaded) ++tolerated;
    }
    if (miss > tolerated) return fail(MRA_ESTATE, std::string("parameters not loaded: ") + names);
  }
  const size_t need = mra_qformer_pair_workspace_bytes(h0, h1, items, L, kv0, kv1);
  if (!workspace || workspace_bytes < need) return fail(MRA_ENOMEM, "workspace too small: need " + std::to_string(need) + " bytes");
  if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MRA_EINVAL, "workspace must be 256-byte aligned");

  hipStream_t stream = as_stream(stream_);
  const int N = items, Q = c.n_query, S = Q + L, H = c.hidden, I = c.inter;
  const int op = h0->op();
  PairWork pw = layout_pair(h0, h1, (char*)workspace, N, L, kv0, kv1);
  Work* w = pw.w;
  const long long SH = (long long)S * H;
  const size_t esz = 2;
  const RowView all_rows = plain(N * S, H), all2 = plain(2 * N * S, H);
  const RowView q_view = items_view(SH, Q, H);
  const RowView t_view = items_view(SH, L > 0 ? L : 1, H);
  const RowView cls_view = items_view(SH, 1, H);
  const RowView qc_rows = plain(N * Q, H), qc2 = plain(2 * N * Q, H);
  int rc;
  bool fold[2];
  for (int l = 0; l < 2; ++l) {
    rc = launch_embed_ln((const long long*)input_ids, N, L, Q, H, hs[l]->cfg.vocab, hs[l]->query, 0, hs[l]->word, hs[l]->pos, hs[l]->embg, hs[l]->embb, c.ln_eps,
                         w[l].hA32, w[l].hA16, nullptr, op, stream);
    if (rc) return chk(rc, "embed_ln");
    fold[l] = hs[l]->ncross > 0 && use_fold(hs[l], kvs[l]);
    if (fold[l]) {
      if (!fold_kmajor(hs[l])) {
        const int kvp = fold_kvp(kvs[l]);
        rc = launch_transpose_pad(encs[l], w[l].encT, kvs[l], hs[l]->cfg.enc_width, kvp, (long long)kvs[l] * hs[l]->cfg.enc_width, (long long)hs[l]->cfg.enc_width * kvp, N, op, stream);
        if (rc) return chk(rc, "enc transpose");
      }
      if ((rc = mra_qformer_prepare(hs[l], stream_))) return rc;
    } else if (hs[l]->ncross > 0) {
      rc = kv_project(hs[l], encs[l], N, kvs[l], w[l].kv16, stream);
      if (rc) return chk(rc, "kv projection gemm");
    }
  }
  const long long* mask2 = nullptr;
  if (attention_mask) {   // the two lanes share the prompt: their mask rows one after the other for the 2 N-item attention launch
    HIP_TRY(hipMemcpyAsync(pw.mask2, attention_mask, (size_t)N * S * 8, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(pw.mask2 + (size_t)N * S, attention_mask, (size_t)N * S * 8, hipMemcpyDeviceToDevice, stream));
    mask2 = pw.mask2;
  }
  const bool ring = N * S >= 1024;
  for (int i = 0; i < c.layers; ++i) {
    const LayerW* Lw[2] = {&h0->layers[i], &h1->layers[i]};
    const bool last = i == c.layers - 1;
    // 1. Q | K | V of both lanes: two problems, one launch
    {
      GemmProb p[2] = {};
      for (int l = 0; l < 2; ++l) {
        p[l].A = w[l].hA16; p[l].a = all_rows; p[l].W = Lw[l]->wqkv; p[l].bias = Lw[l]->bqkv;
        p[l].C = w[l].qkv16; p[l].c = plain(N * S, 3 * H); p[l].M = N * S; p[l].N = 3 * H; p[l].K = H;
        p[l].tile_cfg = ring ? (((hs[l]->chain_ring & 1) && (3 * H) % 144 == 0) ? 9 : 2) : 0;
      }
      if ((rc = launch_gemm(p, 2, EPI_OP, op, stream))) return chk(rc, "qkv gemm (pair)");
    }
    // 2. self-attention of all 2 N items
    {
      AttnArgs a{};
      a.Q = w[0].qkv16; a.K = w[0].qkv16 + (size_t)H * esz; a.V = w[0].qkv16 + (size_t)2 * H * esz; a.O = w[0].ctx16;
      a.q_item_stride = (long long)S * 3 * H; a.q_ld = 3 * H;
      a.k_item_stride = (long long)S * 3 * H; a.k_head_stride = 64; a.k_ld = 3 * H;
      a.v_item_stride = (long long)S * 3 * H; a.v_head_stride = 64; a.v_ld = 3 * H;
      a.o_item_stride = SH; a.o_ld = H;
      a.mask = mask2; a.mask_ld = S;
      a.items = 2 * N; a.heads = c.heads; a.q_rows = S; a.kv_len = S; a.scale = 0.125f; a.nsplit = 1; a.part = nullptr;
      if ((rc = launch_attention(a, op, stream))) return chk(rc, "self attention (pair)");
    }
    // 3. output projections + residual (two problems), 4. LayerNorm of both lanes in one launch
    {
      GemmProb p[2] = {};
      for (int l = 0; l < 2; ++l) {
        p[l].A = w[l].ctx16; p[l].a = all_rows; p[l].W = Lw[l]->wo; p[l].bias = Lw[l]->bo; p[l].R = w[l].hA32; p[l].r = all_rows;
        p[l].C = w[l].pre32; p[l].c = all_rows; p[l].M = N * S; p[l].N = H; p[l].K = H;
        if ((hs[l]->chain_ring & 4) && ring && H % 96 == 0) p[l].tile_cfg = 11;
      }
      if ((rc = launch_gemm(p, 2, EPI_RES_F32, op, stream))) return chk(rc, "attn out gemm (pair)");
      rc = launch_ln_rows4(w[0].pre32, all2, 2 * N * S, H, Lw[0]->ln1g, Lw[0]->ln1b, nullptr, nullptr, N * S, Lw[1]->ln1g, Lw[1]->ln1b, nullptr, nullptr, 1, 1,
                           c.ln_eps, w[0].hB32, all2, w[0].hB16, all2, op, stream);
      if (rc) return chk(rc, "attn ln (pair)");
    }
    const bool cross = Lw[0]->cross_index >= 0;
    if (cross) {
      // 5. cross query projections (two problems), 6. each lane's own cross-attention, 7. output projections + residual, 8. LayerNorm
      GemmProb p[2] = {};
      for (int l = 0; l < 2; ++l) {
        p[l].A = w[l].hB16; p[l].a = q_view; p[l].W = Lw[l]->wcq; p[l].bias = Lw[l]->bcq;
        p[l].C = w[l].qc16; p[l].c = qc_rows; p[l].M = N * Q; p[l].N = H; p[l].K = H;
      }
      if ((rc = launch_gemm(p, 2, EPI_OP, op, stream))) return chk(rc, "cross q gemm (pair)");
      for (int l = 0; l < 2; ++l)
        if ((rc = cross_core(hs[l], w[l], *Lw[l], encs[l], N, kvs[l], fold[l], false, false, stream))) return rc;
      GemmProb o[2] = {};
      for (int l = 0; l < 2; ++l) {
        o[l].A = w[l].ctx16; o[l].a = qc_rows; o[l].W = Lw[l]->wco; o[l].bias = Lw[l]->bco; o[l].R = w[l].hB32; o[l].r = q_view;
        o[l].C = w[l].pre32; o[l].c = qc_rows; o[l].M = N * Q; o[l].N = H; o[l].K = H;
      }
      if ((rc = launch_gemm(o, 2, EPI_RES_F32, op, stream))) return chk(rc, "cross out gemm (pair)");
      // pre32 of lane l holds its compact [N*Q][H] rows at its own base: two launches would be needed for one row view; the rows of the two
      // lanes are N*S*H apart, so address them as items of Q rows with an item stride that jumps lanes: view (item = lane): stride N*S*H, rpi N*Q
      const RowView pre_v = items_view((long long)N * S * H, N * Q, H);
      rc = launch_ln_rows4(w[0].pre32, pre_v, 2 * N * Q, H, Lw[0]->lncg, Lw[0]->lncb, nullptr, nullptr, N * Q, Lw[1]->lncg, Lw[1]->lncb, nullptr, nullptr, 1, 1,
                           c.ln_eps, w[0].hC32, qc2, w[0].hC16, qc2, op, stream);
      if (rc) return chk(rc, "cross ln (pair)");
    }
    // 9-14. feed-forwards: per lane a query problem and a text problem -> four problems per launch
    int text_rows = 0;
    RowView tv = t_view;
    if (L > 0) {
      if (!last) text_rows = N * L;
      else if (outc[0] || outc[1]) { text_rows = N; tv = cls_view; }
    }
    const size_t t_off16 = (size_t)Q * H * esz, t_off32 = (size_t)Q * H;
    const int ng = text_rows > 0 ? 4 : 2;
    {
      GemmProb g[4] = {};
      for (int l = 0; l < 2; ++l) {
        GemmProb& q = g[l];                 // query problems first (they decide the tile), then the text problems
        q.A = cross ? (const void*)w[l].hC16 : (const void*)w[l].hB16; q.a = cross ? qc_rows : q_view;
        q.W = Lw[l]->wiq; q.bias = Lw[l]->biq; q.C = w[l].ffn16; q.c = plain(N * Q, I); q.M = N * Q; q.N = I; q.K = H;
        GemmProb& t = g[2 + l];
        t.A = w[l].hB16 + t_off16; t.a = tv; t.W = Lw[l]->wit; t.bias = Lw[l]->bit;
        t.C = w[l].ffn16 + (size_t)N * Q * I * esz; t.c = plain(text_rows, I); t.M = text_rows; t.N = I; t.K = H;
      }
      if ((h0->chain_ring & 2) && N * Q >= 512 && I % 192 == 0) g[0].tile_cfg = 10;
      if ((rc = launch_gemm(g, ng, EPI_GELU_OP, op, stream))) return chk(rc, "ffn up gemm (pair)");
    }
    {
      GemmProb g[4] = {};
      for (int l = 0; l < 2; ++l) {
        GemmProb& q = g[l];
        q.A = w[l].ffn16; q.a = plain(N * Q, I); q.W = Lw[l]->woq; q.bias = Lw[l]->boq;
        q.R = cross ? w[l].hC32 : w[l].hB32; q.r = cross ? qc_rows : q_view;
        q.C = w[l].pre32; q.c = q_view; q.M = N * Q; q.N = H; q.K = I;
        GemmProb& t = g[2 + l];
        t.A = w[l].ffn16 + (size_t)N * Q * I * esz; t.a = plain(text_rows, I); t.W = Lw[l]->wot; t.bias = Lw[l]->bot;
        t.R = w[l].hB32 + t_off32; t.r = tv; t.C = w[l].pre32 + t_off32; t.c = tv; t.M = text_rows; t.N = H; t.K = I;
      }
      g[0].tile_cfg = N * Q >= 512 && I % 128 == 0 ? 6 : 0;
      if ((h0->chain_ring & 4) && N * Q >= 512 && H % 96 == 0) g[0].tile_cfg = 11;
      if ((rc = launch_gemm(g, ng, EPI_RES_F32, op, stream))) return chk(rc, "ffn down gemm (pair)");
    }
    if (!last) {
      if (text_rows == N * L && L > 0) {
        // query / text LayerNorms of both lanes: four parameter sets, one launch over the 2 N S rows
        rc = launch_ln_rows4(w[0].pre32, all2, 2 * N * S, H, Lw[0]->lnqg, Lw[0]->lnqb, Lw[0]->lntg, Lw[0]->lntb, N * S, Lw[1]->lnqg, Lw[1]->lnqb, Lw[1]->lntg,
                             Lw[1]->lntb, S, Q, c.ln_eps, w[0].hA32, all2, w[0].hA16, all2, op, stream);
        if (rc) return chk(rc, "ffn ln (pair)");
      } else {   // L == 0: query rows only (the whole stream)
        rc = launch_ln_rows4(w[0].pre32, all2, 2 * N * S, H, Lw[0]->lnqg, Lw[0]->lnqb, nullptr, nullptr, N * S, Lw[1]->lnqg, Lw[1]->lnqb, nullptr, nullptr, 1, 1,
                             c.ln_eps, w[0].hA32, all2, w[0].hA16, all2, op, stream);
        if (rc) return chk(rc, "ffn query ln (pair)");
      }
    } else {
      for (int l = 0; l < 2; ++l) {   // last layer: LayerNorm straight into the caller's buffers
        if (outq[l]) {
          rc = launch_ln_rows(w[l].pre32, q_view, N * Q, H, Lw[l]->lnqg, Lw[l]->lnqb, c.ln_eps, outq[l], qc_rows, nullptr, qc_rows, op, stream);
          if (rc) return chk(rc, "final query ln (pair)");
        }
        if (outc[l]) {
          rc = launch_ln_rows(w[l].pre32 + t_off32, cls_view, N, H, Lw[l]->lntg, Lw[l]->lntb, c.ln_eps, outc[l], plain(N, H), nullptr, plain(N, H), op, stream);
          if (rc) return chk(rc, "final cls ln (pair)");
        }
      }
    }
  }
  return MRA_OK;
}

int mra_qformer_set_kv_events(mra_qformer* h, void* ev_start, void* ev_stop) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if ((ev_start == nullptr) != (ev_stop == nullptr)) return fail(MRA_EINVAL, "give both events or neither");
  h->kv_ev0 = reinterpret_cast<hipEvent_t>(ev_start);
  h->kv_ev1 = reinterpret_cast<hipEvent_t>(ev_stop);
  return MRA_OK;
}

int mra_qformer_prepare(mra_qformer* h, void* stream) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (h->ncross == 0) return MRA_OK;
  const mra_cfg& c = h->cfg;
  const size_t H = c.hidden, E = c.enc_width, esz = 2;
  if (h->cross_precise && h->precise_stale) {
    // split-precision cross-attention: W_cq as [H][3H] = (hi | hi | lo), W_k as [heads][E][192] = (hi | hi | lo), from the f32 copies
    int ci = 0;
    for (int i = 0; i < c.layers; ++i) {
      if (h->layers[i].cross_index < 0) continue;
      char* base = h->arena_p + (size_t)ci * precise_layer_bytes(h);
      int rc = launch_split_weight(h->layers[i].wcq32, (int)H, (int)H, base, h->op(), as_stream(stream));
      if (!rc) rc = launch_split_key_weight(h->wk32 + (size_t)ci * H * E, c.heads, (int)E, base + precise_wk_off(h), h->op(), as_stream(stream));
      if (rc) return chk(rc, "split-precision weight preparation");
      ++ci;
    }
    h->precise_stale = false;
  }
  if (!h->fold_stale) return MRA_OK;
  for (int ci = 0; ci < h->ncross; ++ci) {   // W_k [H][E] of cross layer ci -> [heads][E][64]
    const int rc = launch_transpose_pad((const char*)h->wkv + (size_t)(ci * 2) * H * E * esz, h->arena_f + (size_t)ci * H * E * esz, 64, (int)E,
                                        64, (long long)64 * E, (long long)E * 64, c.heads, h->op(), as_stream(stream));
    if (rc) return chk(rc, "key weight regroup");
  }
  h->fold_stale = false;
  return MRA_OK;
}

int mra_qformer_set_cross_mode(mra_qformer* h, int32_t mode) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (mode < 0 || mode > 5)
    return fail(MRA_EINVAL, "cross mode must be 0 (automatic), 1 (K/V cache), 2 (folded), 3 (folded, 128x384 loader-wave tiles), 4 (folded, streaming kernels) or 5 (folded, separate rescale pass)");
  h->cross_mode = mode >= 3 ? 2 : mode;
  h->fold_tile = mode == 3 ? 4 : 2;
  h->fold_stream = mode == 4;     // measured 13 % slower than the loader-wave GEMMs + rescale pass: opt-in
  h->inreg_rescale = mode != 5;   // 5: the round-1 form with a rescale pass over P between the two big GEMMs, kept as the measured alternative
  return MRA_OK;
}

int mra_qformer_set_cross_precision(mra_qformer* h, int32_t mode) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (mode != 0 && mode != 1) return fail(MRA_EINVAL, "cross precision must be 0 (operand dtype) or 1 (split: hi + lo pairs along the score chain)");
  if (mode == 1) {
    const mra_cfg& c = h->cfg;
    if (h->ncross == 0) return fail(MRA_EINVAL, "no cross-attention layers");
    if (c.heads * c.n_query != 384 || h->sc_tile != 5 || !h->split_softmax)
      return fail(MRA_EINVAL, "split precision needs heads * n_query == 384 (the 176 x 384 scores tile)");
    if (!h->arena_p) {
      int dev = 0;
      HIP_TRY(hipGetDevice(&dev));
      if (dev != h->device) return fail(MRA_EINVAL, "handle belongs to another device");
      const hipError_t e = hipMalloc((void**)&h->arena_p, (size_t)h->ncross * precise_layer_bytes(h));
      if (e != hipSuccess) return fail(MRA_ENOMEM, std::string("split-precision weight arena: ") + hipGetErrorString(e));
      h->precise_stale = true;
    }
  }
  h->cross_precise = mode;
  return MRA_OK;
}

int mra_qformer_set_option(mra_qformer* h, const char* name, int32_t value) {
  if (!h || !name) return fail(MRA_EINVAL, "null argument");
  const std::string key(name);
  if (key == "train_ring") {
    if (value < 0 || value > 7) return fail(MRA_EINVAL, "train_ring is a mask of bits 0-2");
    h->train_ring = value;
    return MRA_OK;
  }
  if (key == "chain_ring") {
    if (value < 0 || value > 15) return fail(MRA_EINVAL, "chain_ring is a mask of bits 0-3");
    h->chain_ring = value;
    return MRA_OK;
  }
  return fail(MRA_ENAME, "unknown option: " + key);
}

int mra_qformer_set_kv_done_event(mra_qformer* h, void* ev) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  h->kv_done = reinterpret_cast<hipEvent_t>(ev);
  return MRA_OK;
}

size_t mra_kv_cache_bytes(mra_qformer* h, int32_t items, int32_t kv) {
  if (!h || items <= 0 || kv <= 0) return 0;
  return (size_t)h->ncross * 2 * items * kv * h->cfg.hidden * 2;
}

int mra_kv_project(mra_qformer* h, const void* enc, int32_t items, int32_t kv, void* kv_cache, void* stream) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (items < 0 || kv < 0) return fail(MRA_EINVAL, "negative size");
  if (items == 0 || kv == 0 || h->ncross == 0) return MRA_OK;
  if (!enc || !kv_cache) return fail(MRA_EINVAL, "null argument");
  if ((long long)items * kv > 0x7fffffffLL) return fail(MRA_EINVAL, "items * kv exceeds int32");
  return chk(kv_project(h, enc, items, kv, kv_cache, as_stream(stream)), "kv projection gemm");
}

int mra_llm_proj(mra_qformer* h, const float* z, int32_t rows, void* out, int32_t out_dtype, void* workspace,
                 size_t workspace_bytes, void* stream_) {
  if (!h) return fail(MRA_EINVAL, "null handle");
  if (rows < 0) return fail(MRA_EINVAL, "negative rows");
  if (rows == 0) return MRA_OK;
  const mra_cfg& c = h->cfg;
  if (c.llm_hidden <= 0 || !h->params["llm_proj.weight"].loaded || !h->params["llm_proj.bias"].loaded)
    return fail(MRA_ESTATE, "llm_proj not loaded");
  if (!z || !out || !workspace) return fail(MRA_EINVAL, "null argument");
  const size_t need = align_up((size_t)rows * c.hidden * 2);
  if (workspace_bytes < need) return fail(MRA_ENOMEM, "workspace too small: need " + std::to_string(need));
  const int op = h->op();
  if (out_dtype != MRA_F32 && out_dtype != c.op_dtype) return fail(MRA_EINVAL, "out_dtype must be f32 or the operand dtype");
  hipStream_t stream = as_stream(stream_);
  int rc = launch_convert(z, MRA_F32, workspace, c.op_dtype, (long long)rows * c.hidden, stream);
  if (rc) return chk(rc, "convert z");
  GemmProb p{};
  p.A = workspace; p.a = plain(rows, c.hidden);
  p.W = h->wllm; p.bias = h->bllm;
  p.C = out; p.c = plain(rows, c.llm_hidden);
  p.M = rows; p.N = c.llm_hidden; p.K = c.hidden;
  return chk(launch_gemm(&p, 1, out_dtype == MRA_F32 ? EPI_F32 : EPI_OP, op, stream), "llm_proj gemm");
}

int mra_cosine_score(const float* z, const float* t, int32_t t_rows, int32_t items, int32_t n_query, int32_t hidden,
                     float* sim, float* logit, void* stream) {
  if (items < 0) return fail(MRA_EINVAL, "negative items");
  if (items == 0) return MRA_OK;
  if (!z || !t || !logit) return fail(MRA_EINVAL, "null argument");
  return chk(launch_cosine_score(z, t, t_rows, items, n_query, hidden, 1e-8f, sim, logit, as_stream(stream)), "cosine_score");
}

int mra_fuse_logits(const float* const* logits, const float* weights, int32_t nmod, int32_t n, float* out, void* stream) {
  if (n < 0) return fail(MRA_EINVAL, "negative n");
  if (!logits || !out) return fail(MRA_EINVAL, "null argument");
  for (int m = 0; m < nmod && m < 4; ++m)
    if (!logits[m]) return fail(MRA_EINVAL, "null logits pointer");
  return chk(launch_fuse_logits(logits, weights, nmod, n, out, as_stream(stream)), "fuse_logits");
}

int mra_span_from_logits(const float* logits, int32_t videos, int32_t clips, float alpha, int32_t* spans, void* stream) {
  if (videos < 0) return fail(MRA_EINVAL, "negative videos");
  if (videos == 0) return MRA_OK;
  if (!logits || !spans) return fail(MRA_EINVAL, "null argument");
  return chk(launch_span(logits, videos, clips, alpha, spans, as_stream(stream)), "span_from_logits");
}

}  // extern "C"
