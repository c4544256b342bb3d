/* mra.h -- C ABI of the MI355X (gfx950) cross-modal encode/fuse/score hot path of mrAudio.
 *
 * The reference (globc/mrAudio) is pure Python and has no FFI; the boundary this library stands
 * behind is the duck-typed seam inside XInstructBLIP.generate()/forward():
 *     ln(encoder(frame))                                   models/xinstructblip.py:265,274,822-828
 *     cat(embeds)[indices]                                 models/xinstructblip.py:281-285
 *     {modality}_Qformer.bert(input_ids, attention_mask=, query_embeds=,
 *                             encoder_hidden_states=, encoder_attention_mask=)   :286-293
 *     {modality}_llm_proj(last_hidden_state[:, :32, :])    models/xinstructblip.py:303
 * plus the similarity scorer the north star adds in place of the LLM decode (no reference site).
 * Each entry point below names the reference call it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer on the handle's GPU;
 *   - the caller (PyTorch) owns every buffer, including the workspace; after create/load the
 *     library never allocates device memory and never synchronises the stream;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *   - return 0 on success, a negative MRA_E* code otherwise; mra_last_error() gives the text;
 *   - a handle is bound to the device that was current at create; calls on one handle are not
 *     re-entrant, distinct handles are independent (one per rank / modality).  One exception: after
 *     mra_qformer_prepare, several mra_qformer_forward calls of one handle may be in flight on different
 *     streams (they only read the handle; each needs its own workspace and outputs);
 *   - tensors are dense row-major; matrices of nn.Linear are [out, in] as PyTorch stores them.
 */
#ifndef MRA_H_
#define MRA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRA_OK 0
#define MRA_EINVAL (-1)    /* bad argument / unsupported shape */
#define MRA_ESTATE (-2)    /* weights missing, handle not finalised ... */
#define MRA_EHIP (-3)      /* a HIP call failed */
#define MRA_ENOMEM (-4)    /* workspace too small / allocation failed */
#define MRA_ENAME (-5)     /* unknown weight name */

/* element types */
#define MRA_F32 0
#define MRA_F16 1
#define MRA_BF16 2

typedef struct mra_qformer mra_qformer;

/* Shape of one modality Q-Former; defaults = reference models/xinstructblip.py:614-627
 * (bert-base-uncased + cross-attention every 2nd layer, 32 queries, vocab 30522 + [DEC]). */
typedef struct mra_cfg {
  int32_t hidden;      /* 768; multiple of 256, <= 1024 */
  int32_t heads;       /* 12; hidden / heads must be 64 */
  int32_t inter;       /* 3072 */
  int32_t layers;      /* 12 */
  int32_t cross_freq;  /* 2: layers i % cross_freq == 0 carry cross-attention */
  int32_t enc_width;   /* 1408 (EVA ViT-g) / 768 (BEATs); multiple of 64 */
  int32_t n_query;     /* 32 (fixed by the kernels) */
  int32_t vocab;       /* 30523 */
  int32_t max_pos;     /* 512 */
  float ln_eps;        /* 1e-12 (BERT LayerNorm) */
  float enc_ln_eps;    /* 1e-5  (modality LayerNorm, torch default) */
  int32_t llm_hidden;  /* 4096, or 0 when no llm_proj is loaded */
  int32_t op_dtype;    /* MRA_F16 (default, the reference's autocast dtype) or MRA_BF16: MFMA operand type */
} mra_cfg;

/* Fills *cfg with the reference defaults for the given encoder width. */
void mra_cfg_default(mra_cfg* cfg, int32_t enc_width);

const char* mra_last_error(void);
const char* mra_version(void);

/* ---- lifetime -------------------------------------------------------------------------------
 * replaces: XInstructBLIP.init_Qformer / init_ln / init_vicuna_projection
 * (models/xinstructblip.py:614-655,678-735): construction + weight loading. */
int mra_qformer_create(const mra_cfg* cfg, mra_qformer** out);
void mra_qformer_destroy(mra_qformer* h);

/* Copies (and converts to the operand dtype) one parameter.  `name` uses the LAVIS checkpoint keys
 * with the leading "{modality}_Qformer." stripped, as the reference's loader does
 * (models/xinstructblip.py:647-652): "bert.embeddings.word_embeddings.weight",
 * "bert.encoder.layer.3.attention.self.query.weight", "...crossattention.output.LayerNorm.bias",
 * "...intermediate_query.dense.weight" ... plus "query_tokens" ({m}_query_tokens),
 * "ln.weight"/"ln.bias" ({m}_ln.*), "llm_proj.weight"/"llm_proj.bias" ({m}_llm_proj.*).
 * src is a device pointer of `dtype` with `ndim` dims `shape`.  Unknown name -> MRA_ENAME. */
int mra_qformer_load(mra_qformer* h, const char* name, const void* src, int32_t dtype, const int64_t* shape,
                     int32_t ndim, void* stream);
/* Number of parameters still missing (0 = ready); names are written comma separated into buf. */
int mra_qformer_missing(mra_qformer* h, char* buf, size_t buflen);

/* ---- A2 + A3: modality LayerNorm fused with the sample-major reorder ---------------------------
 * replaces: ln(encoder(frame)) and torch.cat(embeds)[indices]
 * (models/xinstructblip.py:265,274,281-285,822-828).
 * x [n_src_items, tokens, enc_width] of x_dtype; item_index [items] int64 (NULL = identity) gives,
 * for every output item, the source item; out [items, tokens, enc_width] in the operand dtype. */
int mra_modality_ln(mra_qformer* h, const void* x, int32_t x_dtype, const int64_t* item_index, int32_t items,
                    int32_t tokens, void* out, void* stream);

/* ---- A4: Q-Former forward -----------------------------------------------------------------------
 * replaces: {modality}_Qformer.bert(input_ids.repeat(T,1), attention_mask=..., query_embeds=
 * query_tokens.repeat(T,1,1), encoder_hidden_states=..., encoder_attention_mask=ones)
 * (models/xinstructblip.py:286-293).  The learned query tokens are the loaded "query_tokens"
 * (the reference tiles the same [1,32,H] parameter for every item); encoder_attention_mask is
 * all ones in the reference and is not an input here.
 *   input_ids      [items, L] int64
 *   attention_mask [items, n_query + L] int64 (1 = attend) or NULL = all ones
 *   query_embeds   [query_items, n_query, hidden] f32 with query_items == 1 (broadcast) or == items,
 *                  or NULL = the loaded "query_tokens" parameter
 *   enc            [items, kv, enc_width] operand dtype (output of mra_modality_ln)
 *   out_query      [items, n_query, hidden] f32 = last_hidden_state[:, :32, :]        (required)
 *   out_full       [items, n_query + L, hidden] f32 = last_hidden_state, or NULL
 *   out_cls        [items, hidden] f32 = last_hidden_state[:, 32, :], or NULL (needs L >= 1)
 * When neither out_full nor out_cls is given the last layer's text feed-forward is skipped
 * (its result is never read by the reference either: only [:, :32] is sliced, :303). */
size_t mra_qformer_workspace_bytes(mra_qformer* h, int32_t items, int32_t L, int32_t kv);
int mra_qformer_forward(mra_qformer* h, const int64_t* input_ids, const int64_t* attention_mask,
                        const float* query_embeds, int32_t query_items, const void* enc, int32_t items, int32_t L,
                        int32_t kv, float* out_query, float* out_full, float* out_cls, void* workspace,
                        size_t workspace_bytes, void* stream);

/* The dominant launch of the forward on its own (what bench.py prices against the MFMA roofline):
 * K and V of every cross-attention layer, enc [items*kv, enc_width] x Wkv[n_cross*2*hidden, enc_width]^T
 * + bias, written head-major into kv_cache [n_cross][2][items][heads][kv][64] (operand dtype,
 * mra_kv_cache_bytes).  mra_qformer_forward performs exactly this launch as its second step. */
size_t mra_kv_cache_bytes(mra_qformer* h, int32_t items, int32_t kv);
int mra_kv_project(mra_qformer* h, const void* enc, int32_t items, int32_t kv, void* kv_cache, void* stream);

/* Two Q-Formers of equal shape (hidden / heads / inter / layers / cross_freq / n_query / operand dtype) over the SAME prompt in one launch
 * sequence: replaces the two `{m}_Qformer.bert(...)` calls of one step (models/xinstructblip.py:286-293, once per modality).  The layer
 * chains -- identical in shape for every modality -- run as grouped launches (each chain GEMM takes both lanes' problems, the self-attention
 * cores and LayerNorms take the 2 x items rows as one launch); each lane keeps its own cross-attention (encoder width, Kv and formulation may
 * differ).  A chain launch at ~2 k rows is half fixed cost: the grouped sequence spends 0.7 ms less kernel time per
 * headline step and its folded blocks run undisturbed (0.65 instead of 0.79 ms) -- but two forwards on two streams hide ~0.65 ms of the light
 * modality behind the heavy one's latency-bound phases, which one sequence cannot: measured 6.72-6.87 vs 6.57 ms per step.  Kept as a
 * tested alternative (Python: model.pair_forward = True).  Same arithmetic per lane as mra_qformer_forward
 * (learned query tokens, operand-dtype score chain); outputs: out_query [items, 32, H] and / or out_cls [items, H] per lane (fp32).
 * Workspace: mra_qformer_pair_workspace_bytes. */
size_t mra_qformer_pair_workspace_bytes(mra_qformer* h0, mra_qformer* h1, int32_t items, int32_t L, int32_t kv0, int32_t kv1);
int mra_qformer_forward_pair(mra_qformer* h0, mra_qformer* h1, const int64_t* input_ids, const int64_t* attention_mask, const void* enc0,
                             const void* enc1, int32_t items, int32_t L, int32_t kv0, int32_t kv1, float* out_query0, float* out_cls0,
                             float* out_query1, float* out_cls1, void* workspace, size_t workspace_bytes, void* stream);

/* Optional instrumentation for bench.py: when both events (hipEvent_t passed as void*) are non-NULL,
 * every following mra_qformer_forward records ev_start right before and ev_stop right after its
 * K/V-projection launch, on the launch stream.  (NULL, NULL) switches it off. */
int mra_qformer_set_kv_events(mra_qformer* h, void* ev_start, void* ev_stop);

/* How the six cross-attention layers are computed (HF modeling_instructblip.py:464-515 arithmetic either way):
 *   1  K/V cache: K and V of every cross layer projected up front by one GEMM, then a flash-style core per layer;
 *   2  folded:    per layer S = (Q W_k) enc^T, P = softmax(S / 8), context = (P enc) W_v^T + b_v -- half the flops
 *                 at any Kv (the 32 queries are fewer than the 64 head dimensions).  With 12 heads and enc_width % 176 == 0
 *                 the two big products run on the 176 x 384 loader-wave GEMM tile with the softmax split over the
 *                 176-column tiles: tile statistics in the scores epilogue, row factors exp2(m_tile - m_row) / L from a
 *                 small kernel, applied by the P . enc GEMM to its P~ fragments on their way into the MFMA (P is written
 *                 once and read once); otherwise on 128 x 128 tiles with fp32 score rows;
 *   0  automatic (default): folded from Kv >= 2048;
 *   3  as 2 on the 128 x 384 loader-wave tile (A/B runs: no faster);
 *   4  as 2 on the streaming kernels of fold_stream.hip (f16 only: row operands straight to registers two K steps
 *      ahead, slab ring of four LDS slots, power-of-two tile factors applied in registers, no rescale pass).  Measured
 *      13 % SLOWER than mode 2 -- both forms sit on the per-CU load path, DESIGN.md section 8 -- kept, tested, opt-in.
 *   5  as 2 with the row factors applied by a separate rescale pass over P (the round-1 form; same results bit for bit, one more
 *      read and write of P per layer: kept as the measured alternative).
 * mra_qformer_workspace_bytes follows the mode in force; the training entry points always use the K/V cache. */
int mra_qformer_set_cross_mode(mra_qformer* h, int32_t mode);
/* Precision of the cross-attention SCORE chain (no reference counterpart: the reference runs it in fp16 under autocast,
 * models/xinstructblip.py:58-66; its CPU path in fp32):
 *   0  (default) MFMA operands in the operand dtype: the hidden state, W_cq, Q, W_k and Q' = Q W_k are each rounded to 11 bits.
 *      A peaked softmax amplifies that rounding (dp = p (1 - p) ds, |s| up to ~100 for trained, sharply attending Q-Formers);
 *   1  split: every operand of the chain hidden -> Q -> Q' -> scores is carried as an operand-dtype hi + lo pair (~22 bits with
 *      f16): x.w = [xh | xl | xh].[wh | wh | wl]^T on the ordinary GEMM kernels (K tripled on the two small projections), and the
 *      scores product reads Q' as (hi | lo) rows against the SAME encoder slab twice inside one K loop (K doubled, slab bytes
 *      unchanged).  Forces the folded form at any Kv; needs heads * n_query == 384.  The f32 copies of W_cq / W_k live in the parameter
 *      arena (kept current by mra_qformer_load / _load_flat); the first call with mode 1 allocates the prepared-weight arena
 *      (ncross x (3 H H + 3 H E) operand elements) -- the one allocation after create.  Cost and accuracy: DESIGN.md section 8.
 * mra_qformer_workspace_bytes follows the precision in force. */
int mra_qformer_set_cross_precision(mra_qformer* h, int32_t mode);
/* Per-handle tuning options (no reference counterpart; results are the same to fp32 summation order).
 *   "chain_ring"  mask: which GEMMs of the 12-layer chain run on the ring kernel's exact-fit tiles when the launch has ~1 k rows or more:
 *                 bit 0 QKV (144 x 128), bit 1 FFN-up (192 x 128), bit 2 the residual projections (96 x 64), bit 3 (with bit 2) their LayerNorm inside the same
 *                 launch (the last-arriving column tile of a 64-row block normalises it).  DESIGN.md section 8.
 *   "train_ring"  the same mask (bits 0 and 2) for the GEMMs of mra_qformer_forward_train / mra_qformer_backward; default 4. */
int mra_qformer_set_option(mra_qformer* h, const char* name, int32_t value);
/* Derives what the folded path needs from the loaded weights (W_k of every cross layer regrouped per head) on
 * `stream`, if a load made it stale.  mra_qformer_forward does this itself; a caller that runs SEVERAL forwards of one
 * handle concurrently on different streams calls it once before forking. */
int mra_qformer_prepare(mra_qformer* h, void* stream);

/* Scheduling hook: when `ev` (hipEvent_t as void*) is non-NULL every following mra_qformer_forward records it
 * right after its K/V-projection launch, on the launch stream.  The host side makes the light modality's
 * stream wait for the heavy modality's event, so the chip-filling GEMM runs alone and the two latency-bound
 * layer chains overlap each other instead (mraudio_amd/models/xinstructblip.py: fuse_score).  NULL = off. */
int mra_qformer_set_kv_done_event(mra_qformer* h, void* ev);

/* ---- A5: LLM projection ---------------------------------------------------------------------------
 * replaces: {modality}_llm_proj(last_hidden_state[:, :32, :]) (models/xinstructblip.py:303).
 * z [rows, hidden] f32 -> out [rows, llm_hidden] of out_dtype (MRA_F32 or the operand dtype).
 * workspace: rows * hidden operand elements. */
int mra_llm_proj(mra_qformer* h, const float* z, int32_t rows, void* out, int32_t out_dtype, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- A6: scorer (build-defined; the reference has no scorer) ----------------------------------
 * sim[n][q] = cos(z[n][q][:], t[n or 0][:]); logit[n] = max_q sim[n][q].  sim may be NULL. */
int mra_cosine_score(const float* z, const float* t, int32_t t_rows, int32_t items, int32_t n_query, int32_t hidden,
                     float* sim, float* logit, void* stream);
/* out[i] = sum_m weights[m] * logits[m][i] (weights NULL = 1/nmod), fp32, left to right. */
int mra_fuse_logits(const float* const* logits, const float* weights, int32_t nmod, int32_t n, float* out,
                    void* stream);
/* spans[v] = (start, end) inclusive clip indices grown around the first argmax of
 * logits[v*clips .. (v+1)*clips) while the neighbour >= lo + alpha * (hi - lo). */
int mra_span_from_logits(const float* logits, int32_t videos, int32_t clips, float alpha, int32_t* spans,
                         void* stream);

/* ---- training: forward with an activation tape + backward (BASELINE config 5) --------------------
 * Beyond the reference: its Q-Formers are frozen (models/xinstructblip.py:196-204) and utils/trainer.py
 * only updates LoRA adapters of the LLM; the north star asks for a Q-Former fwd+bwd step.  Checked
 * against torch.autograd over the CPU oracle.
 *   mra_qformer_enable_training   builds the transposed weight copies the data-gradient GEMMs read
 *                                 (allocates once; call again after every weight upload)
 *   mra_qformer_forward_train     as mra_qformer_forward (loaded query tokens, optional out_cls) but
 *                                 every layer keeps its activations in `workspace`, which must stay
 *                                 untouched until the matching mra_qformer_backward
 *   mra_qformer_backward          d_out_query [items, n_query, hidden] and/or d_out_cls [items, hidden]
 *                                 (f32, NULL = zero) -> ADDS parameter gradients into `grads`, a flat f32
 *                                 buffer of mra_qformer_grad_bytes() in which parameter `name` (same names
 *                                 as mra_qformer_load) owns numel floats at mra_qformer_grad_offset().
 *                                 Gradients flow to every Q-Former parameter incl. query_tokens and the
 *                                 embeddings; not to enc / ln.* / llm_proj.* (encoder side frozen). */
size_t mra_qformer_grad_bytes(mra_qformer* h);
/* Optimizer-side fast path: refreshes EVERY bert.* parameter in one launch from a flat f32 master buffer
 * laid out exactly like the gradient buffer (parameter `name` at mra_qformer_grad_offset(name)), converting
 * matrices to the operand dtype on the way.  An optimizer that keeps its master weights in that layout
 * (mraudio_amd.qformer re-points every nn.Parameter at its slice) pays one ~0.3 ms kernel per step instead of
 * ~400 mra_qformer_load calls.  query_tokens / ln.* / llm_proj.* still go through mra_qformer_load.
 * No counterpart in the reference (DDP + torch.optim over autograd parameters, utils/trainer.py:60-69). */
int32_t mra_qformer_load_flat(mra_qformer* h, const float* master, size_t master_bytes, void* stream);
int mra_qformer_grad_offset(mra_qformer* h, const char* name, size_t* offset_bytes, int64_t* numel);
int mra_qformer_enable_training(mra_qformer* h, void* stream);
/* The optimizer step of BASELINE config 5 in ONE pass over the flat buffers (replaces, on the hot path, torch.optim.Adam's
 * multi_tensor_apply + mra_qformer_load_flat + the transposed-copy rebuild; caller: utils/trainer.py:137-140 `scaler.step(optimizer)`):
 * torch.optim.Adam's arithmetic (L2 weight decay added to the gradient, bias correction from `step` >= 1) on master / grad / exp_avg /
 * exp_avg_sq -- four fp32 buffers of mra_qformer_grad_bytes() in the gradient buffer's layout -- and, from the same registers, every device
 * copy the update makes stale: the operand-dtype weights, their transposed training copies, fp32 biases / LayerNorms / embeddings / query
 * tokens.  zero_grad != 0 clears the gradient buffer on the way.  ~30 B per parameter of HBM traffic instead of ~50. */
int mra_qformer_adam_step(mra_qformer* h, float* master, float* grad, float* exp_avg, float* exp_avg_sq, size_t bytes, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad, void* stream);
size_t mra_qformer_train_workspace_bytes(mra_qformer* h, int32_t items, int32_t L, int32_t kv);
int mra_qformer_forward_train(mra_qformer* h, const int64_t* input_ids, const int64_t* attention_mask, const void* enc,
                              int32_t items, int32_t L, int32_t kv, float* out_query, float* out_cls, void* workspace,
                              size_t workspace_bytes, void* stream);
int mra_qformer_backward(mra_qformer* h, const int64_t* input_ids, const int64_t* attention_mask, const void* enc,
                         int32_t items, int32_t L, int32_t kv, const float* d_out_query, const float* d_out_cls,
                         float* grads, void* workspace, size_t workspace_bytes, void* stream);

/* ---- introspection for the bench ------------------------------------------------------------------
 * Algorithmic flop count of one mra_qformer_forward (2 flops per MAC; formula in DESIGN.md). */
double mra_qformer_flops(mra_qformer* h, int32_t items, int32_t L, int32_t kv, int32_t with_last_text);

/* ---- EVA ViT-g/14 visual encoder (row A1 / N4) -----------------------------------------------------------
 * replaces: self.video_encoder(frame) inside the per-position loop of XInstructBLIP.generate / forward
 * (models/xinstructblip.py:262-266, :412-420), the module create_eva_vit_g(224, 0, False, "fp16") builds (:658-666).
 * One call encodes ALL frames of a step (the reference's T sequential calls at batch B, collapsed).  Parameter names are
 * the state_dict keys of mraudio_amd/models/eva_vit.py (cls_token, pos_embed, patch_embed.{weight,bias},
 * blocks.{i}.{norm1,norm2}.{weight,bias}, blocks.{i}.attn.{qkv.weight,q_bias,v_bias,proj.weight,proj.bias},
 * blocks.{i}.{fc1,fc2}.{weight,bias}); EvaViTg.hf_state_dict / load_hf_state_dict map them to the HF vision tower. */
typedef struct mra_vit mra_vit;
typedef struct mra_vit_cfg {
  int32_t dim;       /* 1408; multiple of 176 and 64 */
  int32_t heads;     /* 16; head dimension dim / heads = 88 (<= 96, multiple of 8) */
  int32_t mlp;       /* 6144 */
  int32_t depth;     /* 39 (LAVIS drops the 40th EVA block) */
  int32_t patch;     /* 14 */
  int32_t img;       /* 224 -> (224 / 14)^2 + 1 = 257 tokens */
  float ln_eps;      /* 1e-6 */
  int32_t op_dtype;  /* MRA_F16 (the reference's precision="fp16") or MRA_BF16: MFMA operand type */
  int32_t residual_dtype;  /* MRA_F32 (default: fp32 residual stream, the accurate choice) or the operand dtype (the reference's own
                            * precision="fp16" semantics: every residual add rounds to 16 bits; half the epilogue and LayerNorm bytes) */
} mra_vit_cfg;
void mra_vit_cfg_default(mra_vit_cfg* cfg);
int mra_vit_create(const mra_vit_cfg* cfg, mra_vit** out);
void mra_vit_destroy(mra_vit* h);
int mra_vit_load(mra_vit* h, const char* name, const void* src, int32_t dtype, const int64_t* shape, int32_t ndim, void* stream);
/* number of parameters not loaded yet (0 = ready) */
int mra_vit_missing(mra_vit* h);
size_t mra_vit_workspace_bytes(mra_vit* h, int32_t frames);
/* frames [n, 3, img, img] (MRA_F32 or MRA_F16, normalised pixels) -> out [n, tokens, dim] in cfg.residual_dtype: the last block's
 * output, no final norm (the reference's separate video_ln, mra_modality_ln, consumes it in place). */
int mra_vit_forward(mra_vit* h, const void* frames, int32_t dtype, int32_t n, void* out, void* workspace, size_t workspace_bytes,
                    void* stream);
/* algorithmic flops of one forward over `frames` frames (2 per MAC) */
double mra_vit_flops(mra_vit* h, int32_t frames);
/* Per-handle options.  "ln_fold" (default 1; dim = 256 k + 128 only, otherwise ignored): the two LayerNorms of a
 * block (HF modeling_instructblip.py:392-440 layer_norm1 / layer_norm2, the callee of /root/reference/models/xinstructblip.py:262-266) are folded
 * into the GEMMs around them -- LN(x) W^T + b = rstd (x (W diag(g))^T - mu colsum(W diag(g))) + (b + W beta): the residual GEMM before a
 * LayerNorm also writes the rows' operand-dtype copy and 128-column statistics (with the residual stream in the operand dtype the stream is
 * that copy and only 64-column statistics are added), the QKV / fc1 GEMM finishes the LayerNorm in its epilogue; the rows are not read back and
 * no LayerNorm kernel runs.  0: separate LayerNorm launches (round 2's form), for A/B and parity runs.
 * "gemm_persist" (default 1): the QKV / fc1 GEMMs as ONE persistent workgroup per CU walking over the output tiles in the launch's tile order
 * instead of one workgroup per tile (no workgroup turnaround, arguments fetched once; bit-identical results; 0 = off, 2 = only up to 64 rounds).
 * "attn_persist" (default 0; 257-token frames): 1 = the attention core as one persistent workgroup per CU that fetches the K / V rows of its
 * next (frame, head) unit into registers while it works on the current one (bit-identical results; measured slower). */
int mra_vit_set_option(mra_vit* h, const char* name, int32_t value);

/* ---- diagnostics (no reference counterpart) ---------------------------------------------------------------
 * Number of GEMM launches of one main loop ("family") with one epilogue since the library was loaded; read-only, the only
 * process-wide state of the library.  Families (csrc/kernels.h GemmFamily): 0 / 1 two-buffer 64x64 / 128x128, 3 loader-wave 256x256,
 * 4 eight-phase 256x256, 5 / 6 loader-wave 128x384 / 176x384 (folded cross-attention), 7 / 10 128-deep 64x128 / 64x64,
 * 8 eight-phase 128x512 tail tile, 9 eight-phase full + tail tiles in one launch.  Epilogues (GemmEpi): 0 op-dtype, 1 GELU,
 * 2 fp32 residual, 3 fp32, 4 K/V cache, 5 softmax partials, 6 / 7 GELU forward + tape / GELU backward, 8 op-dtype residual.
 * Returns -1 for an unknown family / epilogue.  Used by the parity tests to state which kernel produced the numbers checked. */
int64_t mra_debug_gemm_launches(int32_t family, int32_t epilogue);

#ifdef __cplusplus
}
#endif
#endif /* MRA_H_ */
