"""BASELINE config 5 (``-m gpu``): Q-Former forward + backward on the HIP path against torch.autograd over
the CPU oracle.  The reference never trains its Q-Formers (frozen, models/xinstructblip.py:196-204), so
the oracle's autograd IS the definition of correct here.

Tolerance: both passes feed f16 operands to the MFMAs with fp32 accumulation; a gradient tensor must
agree with the fp32 oracle to 2e-2 in relative Frobenius norm (measured ~3e-3) and no element may be off
by more than 5 % of the tensor's largest entry.
"""
import pytest
import torch

from oracle import qformer_ref as O
from tools.make_golden import make_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _setup(dev, enc_width, seed, op_dtype=torch.float16):
    from mraudio_amd.qformer import QFormer, QFormerConfig, draw_seeded

    cfg = QFormerConfig(enc_width=enc_width, op_dtype=op_dtype)
    qf = QFormer(cfg, device=dev)
    g = qf.init_seeded_(seed=seed, perturb=True)
    qt = draw_seeded(g, (1, cfg.n_query, cfg.hidden), "w", True)
    qf.push("query_tokens", qt)
    qf.push("ln.weight", draw_seeded(g, (enc_width,), "g", True))
    qf.push("ln.bias", draw_seeded(g, (enc_width,), "z", True))
    ocfg = O.QFormerCfg(enc_width=enc_width)
    w = O.init_weights(ocfg, seed=seed, perturb=True)
    return qf, cfg, ocfg, w


# (operand dtype, forward |d| bar, relative-Frobenius bar per gradient tensor, peak-error bar).  bf16 is BASELINE config 5's
# stated dtype: 8 significand bits instead of f16's 11, so every bar is 8x the f16 one (the measured errors scale the same way)
DTYPES = [(torch.float16, 1e-2, 2e-2, 5e-2), (torch.bfloat16, 8e-2, 1.6e-1, 4e-1)]


@pytest.mark.parametrize("op_dtype,fwd_tol,rel_tol,peak_tol", DTYPES, ids=["f16", "bf16"])
def test_forward_backward_matches_oracle_autograd(dev, op_dtype, fwd_tol, rel_tol, peak_tol):
    qf, cfg, ocfg, w = _setup(dev, 1408, 0, op_dtype)
    n, L, kv = 4, 9, 40
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 77, True)
    enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]).to(op_dtype)   # the exact operand the kernels see
    g = torch.Generator().manual_seed(5)
    rq, rc = torch.randn(n, 32, 768, generator=g), torch.randn(n, 768, generator=g)

    # oracle: fp32 autograd
    wl = {k: (v.clone().requires_grad_(True) if k.startswith("bert.") or k == "query_tokens" else v) for k, v in w.items()}
    h = O.qformer_forward(wl, ocfg, ids, att, wl["query_tokens"].expand(n, -1, -1), enc.float())
    loss_ref = (h[:, :32] * rq).sum() + (h[:, 32] * rc).sum()
    loss_ref.backward()

    # HIP path
    q, c = qf.forward_train(ids.to(dev), att.to(dev), enc.to(dev))
    assert (q.cpu() - h[:, :32].detach()).abs().max().item() < fwd_tol
    assert (c.cpu() - h[:, 32].detach()).abs().max().item() < fwd_tol
    inf = qf.forward_fused(ids.to(dev), att.to(dev), enc.to(dev), want_query=True, want_cls=True)
    assert (inf["query"] - q).abs().max().item() < fwd_tol          # training keeps the pre-GELU value in f16 (one more rounding)
    loss = (q * rq.to(dev)).sum() + (c * rc.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()

    worst, worst_name = 0.0, ""
    for k, ref in wl.items():
        if not (k.startswith("bert.") or k == "query_tokens"):
            continue
        gref = ref.grad
        got = qf.grad_of(k).view_as(gref).cpu()
        if k == "bert.embeddings.word_embeddings.weight":
            assert torch.count_nonzero(got).item() > 0
        denom = gref.norm().item()
        if k.endswith("key.bias"):
            # softmax is invariant to a per-query shift of the scores, so the true key-bias gradient is 0 (the
            # oracle's value is fp32 noise): the kernel's must be noise too, measured against the query-bias one
            scale = wl[k.replace(".key.", ".query.")].grad.norm().item()
            assert got.norm().item() < rel_tol * scale + 1e-4, (k, got.norm().item(), scale)
            continue
        rel = ((got - gref).norm() / denom).item()
        peak = (got - gref).abs().max().item() / gref.abs().max().item()
        if rel > worst:
            worst, worst_name = rel, k
        assert rel < rel_tol and peak < peak_tol, (k, rel, peak)
    print("worst relative gradient error", op_dtype, worst, worst_name)
    # parameter .grad fields are views of the flat buffer
    p = qf.bert.encoder.layer[3].attention.output.dense.weight
    assert p.grad is not None and p.grad.data_ptr() == qf.grad_of("bert.encoder.layer.3.attention.output.dense.weight").data_ptr()


def test_gradient_accumulation_and_zeroing(dev):
    qf, cfg, ocfg, w = _setup(dev, 768, 1)
    n, L, kv = 2, 5, 24
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 3, False)
    enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]).half().to(dev)
    key = "bert.encoder.layer.0.crossattention.self.value.weight"

    def run():
        q, c = qf.forward_train(ids.to(dev), att.to(dev), enc)
        (q.sum() + c.sum()).backward()

    run()
    g1 = qf.grad_of(key).clone()
    run()                                            # second micro-batch accumulates (utils/trainer.py:31,137 grad-accum 2)
    assert torch.allclose(qf.grad_of(key), 2 * g1, rtol=1e-3, atol=1e-5)
    for p in qf.bert.parameters():                   # what optimizer.zero_grad(set_to_none=True) does
        p.grad = None
    run()
    assert torch.allclose(qf.grad_of(key), g1, rtol=1e-3, atol=1e-5)


def test_one_training_step_moves_the_loss(dev):
    """fwd + bwd + Adam on the Q-Former parameters: the loss of the same batch must drop."""
    qf, cfg, ocfg, w = _setup(dev, 768, 2)
    n, L, kv = 4, 6, 32
    ids, tmask, att, feats = make_inputs(ocfg, n, L, kv, 9, False)
    enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]).half().to(dev)
    target = torch.randn(n, 32, 768, generator=torch.Generator().manual_seed(1)).to(dev)
    qf.enable_training()
    params = [p for k, p in qf.bert.named_parameters() if "word_embeddings" not in k]
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)   # fused steps do not bump version counters: the re-upload must not rely on them
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        q, _ = qf.forward_train(ids.to(dev), att.to(dev), enc, want_cls=False)
        loss = torch.nn.functional.mse_loss(q, target)
        loss.backward()
        opt.step()                                   # in-place update; the next forward re-uploads the weights
        losses.append(loss.item())
    assert losses[2] < losses[0], losses


def test_model_level_finetune_step(dev):
    """finetune.py-shaped use (utils/trainer.py:124-140): loss = model(samples)["loss"]; loss.backward();
    optimizer.step() -- on a Charades-STA-shaped synthetic batch (B = 1, T = 20), both modalities."""
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    model = XInstructBLIP(seed=3, perturb=True, device=dev)
    model.enable_qformer_training()
    g = torch.Generator().manual_seed(2)
    samples = {"video_embeds": torch.randn(1, 20, 257, 1408, generator=g), "audio_embeds": torch.randn(1, 20, 256, 768, generator=g),
               "text_input": ["Query: a person opens the door.\nGiven the video and the query, find the relevant windows.\nRelevant windows: "],
               "text_output": ["[[6, 12]]"], "timestamps": [list(range(0, 40, 2))], "duration": [40]}
    params = [p for d in model.get_optimizer_params(0.05) for p in d["params"]]
    assert len(params) > 500 and all(p.requires_grad for p in params)
    params = [p for p in params if p.shape[0] != 30523]   # leave the word embeddings out of Adam (sparse rows)
    opt = torch.optim.SGD(params, lr=1.0)
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = model(samples)["loss"]
        loss.backward()
        model.all_reduce_grads()
        assert model.video_query_tokens.grad is not None and torch.isfinite(model.video_query_tokens.grad).all()
        assert torch.isfinite(model.audio_Qformer._grad_flat).all()
        # a fixed-length step along the negative gradient (length 0.05 in parameter space): if the gradients are
        # right the loss of the same batch must go down to first order
        gnorm = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params)).item()
        assert gnorm > 0
        for grp in opt.param_groups:
            grp["lr"] = 0.05 / gnorm
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] and losses[1] < losses[0], losses
    # inference after training uses the updated weights
    out = model.generate(samples)
    assert len(out) == 1 and out[0].startswith("[[")


def test_modalities_on_their_own_streams_give_the_same_loss_and_gradients(dev):
    """``model.train_streams``: each modality's forward (and therefore its backward) on its own stream -- an A/B switch of
    ``tools/bench_finetune.py --streams`` (no gain measured: DESIGN.md section 8).  Same loss, same gradients up to the summation
    order of the weight-gradient atomics."""
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    model = XInstructBLIP(seed=3, perturb=True, device=dev)
    model.enable_qformer_training()
    g = torch.Generator().manual_seed(5)
    samples = {"video_embeds": torch.randn(1, 6, 257, 1408, generator=g), "audio_embeds": torch.randn(1, 6, 256, 768, generator=g),
               "text_input": ["Query: a person opens the door.\nGiven the video and the query, find the relevant windows.\nRelevant windows: "],
               "text_output": ["[[2, 6]]"], "timestamps": [list(range(0, 12, 2))], "duration": [12]}
    got = {}
    for streams in (False, True):
        model.train_streams = streams
        for m in ("video", "audio"):
            getattr(model, f"{m}_Qformer")._grad_flat.zero_()
        loss = model(samples)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        got[streams] = (loss.item(), {m: getattr(model, f"{m}_Qformer")._grad_flat.clone() for m in ("video", "audio")})
    assert abs(got[True][0] - got[False][0]) <= 1e-6 * max(1.0, abs(got[False][0]))
    for m in ("video", "audio"):
        a, b = got[False][1][m], got[True][1][m]
        assert torch.isfinite(a).all() and a.abs().max().item() > 0
        assert ((a - b).norm() / a.norm()).item() < 1e-4, m


def test_flat_parameter_update_equals_per_tensor_update(dev):
    """Optimizer steps on the flat parameter (master buffer + gradient buffer) must move every per-tensor view as
    per-tensor steps would; zero_grad on the flat parameter restarts the accumulation.  Momentum SGD, not Adam:
    Adam's g / (|g| + eps) turns the last-bit noise of the atomically summed weight gradients into sign flips."""
    from mraudio_amd.models.xinstructblip import XInstructBLIP

    g = torch.Generator().manual_seed(2)
    samples = {"video_embeds": torch.randn(1, 6, 257, 1408, generator=g), "audio_embeds": torch.randn(1, 6, 256, 768, generator=g),
               "text_input": ["Query: a person opens the door.\nRelevant windows: "], "text_output": ["[[2, 6]]"],
               "timestamps": [list(range(0, 12, 2))], "duration": [12]}
    results = []
    for flat in (False, True):
        model = XInstructBLIP(seed=3, perturb=True, device=dev)
        model.enable_qformer_training()
        if flat:
            params = model.flat_optimizer_params()
            assert len(params) == 2 and params[0].grad is not None
        else:
            params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=0.01, momentum=0.9)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            model(samples)["loss"].backward()
            opt.step()
        w = model.video_Qformer.bert.encoder.layer[4].crossattention.self.query.weight.detach().clone()
        results.append((w, model.video_query_tokens.detach().clone(), float(model(samples)["loss"])))
        assert "_flat_param" not in model.state_dict() and not any("_flat" in k for k, _ in model.named_parameters())
    (w0, q0, l0), (w1, q1, l1) = results
    # two identical per-tensor runs differ by ~2e-5 in the query tokens at lr 0.05 (atomic summation order of the
    # gradients): the flat and the per-tensor update must agree to that noise level, far below one update (~1e-3)
    assert (w0 - w1).abs().max().item() < 1e-5 and (q0 - q1).abs().max().item() < 5e-5
    assert abs(l0 - l1) < 1e-3


def test_fused_adam_step_equals_torch_adam_on_the_same_gradients(dev):
    """``mra_qformer_adam_step`` (update + device-copy refresh + gradient clearing in one pass) against ``torch.optim.Adam`` stepping a
    copy of the same master buffer with the same gradients: three steps, weight decay on, bias correction exercised; then the DEVICE
    copies really are the updated weights (a forward through the handle equals a forward after re-uploading the master values)."""
    qf, cfg, ocfg, w = _setup(dev, 768, 2)
    qf.enable_training()
    n = qf._master_flat.numel()
    ref_p = torch.nn.Parameter(qf._master_flat.detach().clone())
    opt = torch.optim.Adam([ref_p], lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.02)
    g = torch.Generator(device=dev).manual_seed(5)
    for step in range(3):
        grad = torch.randn(n, generator=g, device=dev) * (0.5 ** step)
        qf._grad_flat.copy_(grad)
        ref_p.grad = grad.clone()
        opt.step()
        qf.adam_step(3e-3, (0.9, 0.99), 1e-8, 0.02, zero_grad=True)
        for name in ("bert.encoder.layer.3.output.dense.weight", "bert.embeddings.word_embeddings.weight", "bert.encoder.layer.0.attention.self.key.bias", "query_tokens"):
            assert float(qf.grad_of(name).abs().max()) == 0.0, name               # every parameter's gradient is cleared on the way
        for name in ("bert.encoder.layer.3.output.dense.weight", "bert.embeddings.word_embeddings.weight", "bert.encoder.layer.11.output_query.LayerNorm.bias", "query_tokens"):
            off, numel = qf._slice_of(name)
            d = (qf._master_flat[off: off + numel] - ref_p.detach()[off: off + numel]).abs().max().item()
            assert d < 2e-6, (step, name, d)                                    # same arithmetic, fp32 rounding order only
    # the device copies follow: forward through the handle == forward of a fresh handle loaded with the master values
    ids, tmask, att, feats = make_inputs(ocfg, 3, 5, 40, 9, False)
    enc = O.modality_layernorm(feats, w["ln.weight"], w["ln.bias"]).half().to(dev)
    z_fused = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"].clone()
    q_train, _ = qf.forward_train(ids.to(dev), att.to(dev), enc, want_cls=False)   # uses the transposed copies' siblings (same weights)
    qf._dirty = True                                                             # force the ordinary re-upload path (mra_qformer_load_flat)
    z_reload = qf.forward_fused(ids.to(dev), att.to(dev), enc, want_query=True)["query"]
    assert torch.equal(z_fused, z_reload)
    assert (q_train.detach() - z_fused).abs().max().item() < 2e-2


def test_training_loss_equals_the_oracle_bce_and_one_fused_step_follows_torch_adam(dev):
    """Row N1: ``model(samples)["loss"]`` of a Charades-STA-shaped batch equals the build-defined objective evaluated on
    the ORACLE's logits (binary cross-entropy of sigmoid(20 x fused logit) against clip membership, 1e-3), and one ``FusedQFormerAdam``
    step moves the parameters as ``torch.optim.Adam`` does on the same (HIP-computed) gradients."""
    from mraudio_amd.models.xinstructblip import ENC_WIDTH, XInstructBLIP
    from mraudio_amd.utils.optim import FusedQFormerAdam

    model = XInstructBLIP(seed=3, perturb=True, device=dev)
    model.enable_qformer_training()
    g = torch.Generator().manual_seed(2)
    T = 20
    samples = {"video_embeds": torch.randn(1, T, 257, 1408, generator=g), "audio_embeds": torch.randn(1, T, 256, 768, generator=g),
               "text_input": ["Query: a person opens the door.\nGiven the video and the query, find the relevant windows.\nRelevant windows: "],
               "text_output": ["[[6, 12]]"], "timestamps": [list(range(0, 2 * T, 2))], "duration": [2 * T]}
    loss = model(samples)["loss"]
    # the oracle on the same inputs
    text = model.tokenizer(samples["text_input"], padding="longest", truncation=True, max_length=128, return_tensors="pt")
    ids, tm = text.input_ids.repeat(T, 1), text.attention_mask.repeat(T, 1)
    cfgs = {m: O.QFormerCfg(enc_width=ENC_WIDTH[m]) for m in ("video", "audio")}
    ws = {"video": O.init_weights(cfgs["video"], seed=3, perturb=True), "audio": O.init_weights(cfgs["audio"], seed=4, perturb=True)}
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    with torch.no_grad():
        ref = O.encode_fuse_score(ws, cfgs, {m: samples[f"{m}_embeds"][0] for m in ("audio", "video")}, ids, tm, 1, T)
    ts = torch.tensor(samples["timestamps"][0], dtype=torch.float32)
    target = ((ts >= 6) & (ts <= 12)).float()
    want = torch.nn.functional.binary_cross_entropy_with_logits(ref["fused"] * 20.0, target)
    assert abs(loss.item() - want.item()) <= 1e-3 * max(1.0, abs(want.item())), (loss.item(), want.item())
    # one optimizer step: FusedQFormerAdam vs torch.optim.Adam on copies of the same master values and gradients
    loss.backward()
    refs = []
    for m in model.modalities:
        qf = getattr(model, f"{m}_Qformer")
        p = torch.nn.Parameter(qf._master_flat.detach().clone())
        p.grad = qf._grad_flat.detach().clone()
        refs.append(p)
    ropt = torch.optim.Adam(refs, lr=1e-4)
    ropt.step()
    opt = FusedQFormerAdam(model, lr=1e-4)
    opt.step()
    for m, p in zip(model.modalities, refs):
        qf = getattr(model, f"{m}_Qformer")
        moved = (p.detach() - qf._master_flat).abs().max().item()
        assert moved < 1e-7 + 2e-2 * 1e-4, (m, moved)          # 2e-2 of one update (lr = 1e-4 per element at most)
    opt.zero_grad()
    # the step refreshed the device copies itself: the next forward equals one after a forced re-upload of the master values
    loss2 = model(samples)["loss"]
    for m in model.modalities:
        getattr(model, f"{m}_Qformer")._dirty = True
    loss3 = model(samples)["loss"]
    assert torch.isfinite(loss2) and abs(loss2.item() - loss3.item()) < 1e-6 and abs(loss2.item() - loss.item()) > 1e-6
