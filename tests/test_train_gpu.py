"""
Gradient parity of the training kernels (tape autograd over libtocvp) against torch.autograd on the
CPU in fp64 / fp32: every differentiable op, then the assembled predictor training step against the
oracle differentiated by torch.  Needs a real MI355X (pytest -m gpu).
"""

import pytest
import torch
import torch.nn.functional as F

from textocvp_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(name, shape, dist="normal", scale=1.0):
    return synth.synth_tensor("train." + name, shape, dist, scale)


def rel_err(got, ref):
    ref = ref.double()
    return (got.detach().cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _ag():
    from textocvp_amd.train import autograd
    return autograd


@pytest.mark.parametrize("rows,wgrad", [(112, "fp32"), (1600, "fp32"), (1600, "bf16x3"), (112, "bf16x3")])
def test_linear_backward_transpose_free_weight_gradient(rows, wgrad, monkeypatch):
    """ N, K multiples of 128 and M % 16 == 0: dW and db come from tocvp_gemm_tn_f32 / tocvp_gemm_tn_bf16x3_f32 (split-K
    partial sums kept per weight, every use of the weight inside one backward pass accumulates into them; the splits
    are added once when the tape finishes).  Two uses with different row counts, then a second pass.  The split-bf16
    kernel (three products of two bf16 planes, M % 32 == 0 -- 336 rows fall back to fp32) is held to 16-bit-mantissa
    accuracy, the exact fp32 MFMA to 1e-5. """
    ag = _ag()
    from textocvp_amd import kernels as K
    assert ag._TN
    monkeypatch.setattr(ag, "_WGRAD_PRECISION", wgrad)
    monkeypatch.setattr(ag, "_DGRAD_PRECISION", "bf16x6")
    wtol = 1e-5 if wgrad == "fp32" or rows == 112 else 2e-5
    N, Kd = 256, 384
    w, b = rnd("tw", (N, Kd), "uniform", Kd ** -0.5), rnd("tb", (N,), "uniform", 0.1)
    xs = [rnd("tx0", (3, rows, Kd)), rnd("tx1", (1, 48, 2, Kd))]
    gs = [rnd("tg0", (3, rows, N)) * 1e-4, rnd("tg1", (1, 48, 2, N)) * 1e-4]
    wr, br = w.double().requires_grad_(), b.double().requires_grad_()
    xrs = [x.double().requires_grad_() for x in xs]
    for xr, g in zip(xrs, gs):                         # no activation: a ReLU whose pre-activation is within
        (xr @ wr.t() + br).backward(g.double())        # rounding of 0 would flip against the fp64 reference
    W, B = ag.Var(w.to(DEV), True), ag.Var(b.to(DEV), True)
    for rep in (1, 2):
        tape = ag.Tape()
        Xs = [ag.Var(x.to(DEV), True) for x in xs]
        Ys = [ag.linear(tape, X, W, B) for X in Xs]
        for Y, g in zip(Ys, gs):
            Y.grad = g.to(DEV)
        tape.backward()
        print(f"weight gradient ({wgrad}, {rows} rows): rel err {rel_err(W.grad, rep * wr.grad):.2e}")
        assert rel_err(W.grad, rep * wr.grad) < wtol and rel_err(B.grad, rep * br.grad) < 1e-5
        for X, xr in zip(Xs, xrs):
            assert rel_err(X.grad, xr.grad) < 1e-5
    # deterministic: the same tape twice gives the same bits
    outs = []
    for _ in range(2):
        W2, B2 = ag.Var(w.to(DEV), True), ag.Var(b.to(DEV), True)
        tape = ag.Tape()
        Y = ag.linear(tape, ag.Var(xs[0].to(DEV), True), W2, B2)
        Y.grad = gs[0].to(DEV)
        tape.backward()
        outs.append((W2.grad.clone(), B2.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("single_use", [False, True])
def test_mlp_backward_with_gated_hidden_gradient(single_use):
    """ Linear-ReLU-Linear(+residual): with ``single_use`` the hidden gradient leaves the second linear's
    data-gradient GEMM already masked by the ReLU (TOCVP_ACT_GATE epilogue); same gradients either way.
    The hidden pre-activations are kept away from 0 so that the fp64 reference takes the same branches. """
    ag = _ag()
    from textocvp_amd import kernels as K
    rows, D, Hd = 1536, 128, 256
    x = rnd("mx", (rows, D))
    w1, b1 = rnd("mw1", (Hd, D), "uniform", D ** -0.5), rnd("mb1", (Hd,), "uniform", 0.1)
    w2, b2 = rnd("mw2", (D, Hd), "uniform", Hd ** -0.5), rnd("mb2", (D,), "uniform", 0.1)
    pre = x.double() @ w1.double().t() + b1.double()
    x = x + 0.0
    close = pre.abs().min(dim=1).values < 1e-4            # rows with a pre-activation near the kink: drop them
    x = x[~close][:1024].contiguous()
    gy = rnd("mg", (1024, D))[: x.shape[0]] * 1e-3
    ref = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2)]
    xr, w1r, b1r, w2r, b2r = ref
    (torch.relu(xr @ w1r.t() + b1r) @ w2r.t() + b2r + xr).backward(gy.double())
    tape = ag.Tape()
    X, W1, B1, W2, B2 = (ag.Var(t.to(DEV), True) for t in (x, w1, b1, w2, b2))
    Hv = ag.linear(tape, X, W1, B1, act=K.ACT_RELU)
    Hv.single_use = single_use
    Y = ag.linear(tape, Hv, W2, B2, residual=X)
    Y.grad = gy.to(DEV)
    tape.backward()
    assert Hv.gated == single_use
    for got, r in zip((X, W1, B1, W2, B2), ref):
        assert rel_err(got.grad, r.grad) < 1e-5


@pytest.mark.parametrize("rows", [100, 512, 4096])
@pytest.mark.parametrize("act", ["none", "relu", "gelu"])
def test_linear_backward(act, rows):
    """ rows = 100: generic fp32 batched GEMM; rows = 512 (M = 1536): the bf16x6 split GEMM on transposed
    copies; rows = 4096 (M = 12288): split-K weight gradient (batched partial products + column sum) """
    ag = _ag()
    from textocvp_amd import kernels as K
    N, Kd = 192, 128
    x, w, b = rnd("lx", (3, rows, Kd)), rnd("lw", (N, Kd), "uniform", Kd ** -0.5), rnd("lb", (N,), "uniform", 0.1)
    gy = rnd("lg", (3, rows, N)) * 1e-4                  # small gradients: must not be flushed
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    pre = xr @ wr.t() + br
    y = {"none": pre, "relu": torch.relu(pre), "gelu": F.gelu(pre)}[act]
    y.backward(gy.double())
    tape = ag.Tape()
    X, W, B = (ag.Var(t.to(DEV), True) for t in (x, w, b))
    code = {"none": K.ACT_NONE, "relu": K.ACT_RELU, "gelu": K.ACT_GELU}[act]
    Y = ag.linear(tape, X, W, B, act=code)
    Y.grad = gy.to(DEV)
    tape.backward()
    assert rel_err(Y.data, y) < 1e-5
    assert rel_err(X.grad, xr.grad) < 1e-5
    assert rel_err(W.grad, wr.grad) < 1e-5
    assert rel_err(B.grad, br.grad) < 1e-5
    # a second use of the same weight accumulates into the existing gradient
    tape = ag.Tape()
    Y2 = ag.linear(tape, X, W, B, act=code)
    Y2.grad = gy.to(DEV)
    tape.backward()
    assert rel_err(W.grad, 2 * wr.grad) < 1e-5 and rel_err(B.grad, 2 * br.grad) < 1e-5


def test_layer_norm_backward():
    ag = _ag()
    x = rnd("nx", (700, 512))
    g, b = 1 + rnd("ng", (512,), "uniform", 0.2), rnd("nb", (512,), "uniform", 0.1)
    gy = rnd("ngy", (700, 512))
    xr, gr, br = (t.double().requires_grad_() for t in (x, g, b))
    F.layer_norm(xr, (512,), gr, br, 1e-6).backward(gy.double())
    tape = ag.Tape()
    X, G, B = (ag.Var(t.to(DEV), True) for t in (x, g, b))
    Y = ag.layer_norm(tape, X, G, B, 1e-6)
    Y.grad = gy.to(DEV)
    tape.backward()
    assert rel_err(X.grad, xr.grad) < 2e-5
    assert rel_err(G.grad, gr.grad) < 2e-5
    assert rel_err(B.grad, br.grad) < 2e-5


@pytest.mark.parametrize("H,E", [(4, 128), (4, 256)])
@pytest.mark.parametrize("Tq,Tk,lens", [(70, 70, None), (30, 12, [12, 5, 9]), (300, 300, None), (150, 40, [40, 33, 1])])
def test_attention_backward(Tq, Tk, lens, H, E):
    """ head size 32: the batched-GEMM chain (scores materialised); head size 64 (the predictor's): the fused
    backward of csrc/attn_bwd.hip, ragged tile edges and key padding included """
    ag = _ag()
    B = 3
    q, k, v = rnd("aq", (B, Tq, E)), rnd("ak", (B, Tk, E)), rnd("av", (B, Tk, E))
    go = rnd("ago", (B, Tq, E))
    scale = (E // H) ** -0.5
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))

    def heads(t, T):
        return t.reshape(B, T, H, E // H).transpose(1, 2)
    s = heads(qr, Tq) @ heads(kr, Tk).transpose(-1, -2) * scale
    if lens is not None:
        mask = torch.arange(Tk)[None, :] >= torch.tensor(lens)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, -1) @ heads(vr, Tk)).transpose(1, 2).reshape(B, Tq, E)
    o.backward(go.double())
    tape = ag.Tape()
    Q, Kv, V = (ag.Var(t.to(DEV), True) for t in (q, k, v))
    kl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=DEV)
    O = ag.attention(tape, Q, Kv, V, H, scale, key_len=kl)
    O.grad = go.to(DEV)
    tape.backward()
    assert rel_err(O.data, o) < 1e-5
    for got, ref in ((Q, qr), (Kv, kr), (V, vr)):
        assert rel_err(got.grad, ref.grad) < 2e-5


def test_position_rows_embedding_mse_and_accumulation():
    ag = _ag()
    B, w, Ks, E = 2, 3, 5, 64
    x, pe = rnd("px", (B, w, Ks, E)), rnd("pp", (10, E))
    tgt = rnd("pt", (B, w, Ks, E))
    index = [9, 8, 7]
    xr, per = x.double().requires_grad_(), pe.double().requires_grad_()
    y = xr + per[index][None, :, None, :]
    z = y + y * 0 + xr                                   # x used twice: gradient accumulation
    loss = 0.7 * F.mse_loss(z, tgt.double())
    loss.backward()
    tape = ag.Tape()
    X, PE = ag.Var(x.to(DEV), True), ag.Var(pe.to(DEV), True)
    Y = ag.add_position_rows(tape, X, PE, index)
    Z = ag.add(tape, Y, X)
    total, sc = ag.mse(tape, Z, tgt.to(DEV), weight=0.7)
    tape.backward()
    assert abs(total.item() * sc - loss.item()) < 1e-5 * abs(loss.item())
    assert rel_err(X.grad, xr.grad) < 1e-5
    assert rel_err(PE.grad, per.grad) < 1e-5
    # embedding scatter-add with repeated ids
    ids = torch.tensor([[1, 4, 4, 0], [2, 1, 1, 1]])
    tab = rnd("et", (6, 32))
    ge = rnd("eg", (2, 4, 32))
    tr = tab.double().requires_grad_()
    F.embedding(ids, tr).backward(ge.double())
    tape = ag.Tape()
    T = ag.Var(tab.to(DEV), True)
    Eo = ag.embedding(tape, ids.to(DEV), T)
    Eo.grad = ge.to(DEV)
    tape.backward()
    assert rel_err(T.grad, tr.grad) < 1e-5


def _predictor_setup(num_slots=7, num_preds=3, B=2):
    from textocvp_amd.setup_model import default_exp_params, setup_predictor
    exp = default_exp_params(num_slots=num_slots, num_context=1, num_preds=num_preds)
    pred = setup_predictor(exp)
    synth.fill_module_(pred, prefix="pred.")
    hist = synth.synth_tensor("train.hist", (B, 1 + num_preds, num_slots, 128), "normal")
    tokens, lengths = synth.synth_captions(B, max_len=12, lengths=[9, 12][:B], seed=0)
    return exp, pred, hist, tokens, lengths


def test_predictor_rollout_gradients_match_oracle_autograd():
    """ slot-MSE loss through the 3-step autoregressive rollout (BPTT, text encoder included): every
    parameter gradient of the tape autograd equals torch.autograd applied to the CPU oracle (fp64) """
    from oracle import slot_rollout_oracle as O
    from textocvp_amd.train import autograd as ag
    from textocvp_amd.train.predictor import TrainablePredictor
    exp, pred, hist, tokens, lengths = _predictor_setup()
    P = 3
    sd = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in pred.state_dict().items()}
    ref_preds = O.rollout(sd, hist.double(), tokens, lengths, 1, P)
    target = hist[:, 1:1 + P]
    ref_loss = F.mse_loss(ref_preds, target.double())
    ref_loss.backward()

    pred = pred.to(DEV)
    tp = TrainablePredictor(pred, text_dropout=0.0)
    tape = ag.Tape()
    preds = tp.rollout(tape, hist.to(DEV), tokens.to(DEV), lengths.to(DEV), P)
    stacked = ag.stack_frames(tape, preds)
    total, sc = ag.mse(tape, stacked, target.to(DEV))
    tape.backward()
    assert abs(total.item() * sc - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    assert rel_err(stacked.data, ref_preds) < 1e-4
    worst = 0.0
    for name, var in tp.names.items():
        ref = sd[name].grad
        if ref is None:
            assert var.grad is None or var.grad.abs().max().item() == 0.0, name
            continue
        assert var.grad is not None, name
        e = rel_err(var.grad, ref)
        worst = max(worst, e)
        assert e < 2e-3, (name, e)
    print(f"predictor BPTT gradients: worst relative error {worst:.2e} over {len(tp.names)} tensors")


def test_textocvp_t5_rollout_gradients_with_frozen_text_encoder():
    """ TextOCVP_T5 (reference text_cond_OCVP.py:141-151: pretrained T5 encoder, freeze_params): the text
    embeddings come from the frozen encoder (inference kernels, pinned by t5_encoder.npz elsewhere), the predictor
    blocks train.  Gradients of every trainable tensor against torch.autograd on the oracle's predictor step fed
    with the same embeddings; the frozen encoder has neither gradients nor optimiser state. """
    from conftest import load_golden
    from oracle import slot_rollout_oracle as O
    from textocvp_amd.setup_model import default_exp_params, setup_predictor
    from textocvp_amd.train import autograd as ag
    from textocvp_amd.train.predictor import TrainablePredictor
    g = load_golden("t5_encoder.npz")
    exp = default_exp_params(num_slots=7, num_context=1, num_preds=3, predictor_name="TextOCVP_T5")
    pred = setup_predictor(exp)
    synth.fill_module_(pred.predictor.text_encoder, prefix="t5.")
    for part in ("predictor", "mlp_in", "mlp_out", "pe"):
        synth.fill_module_(getattr(pred.predictor, part), prefix=f"pred.predictor.{part}.")
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    B, P = ids.shape[0], 3
    # (the seed "train.hist_t5" of rounds 2-4 puts ONE hidden unit of block 1's MLP on its ReLU kink: the all-fp32 arithmetic
    # gates it the other way than the float64 reference, 1.2e-2 on that one gradient -- a property of the input, not of the
    # kernels; this seed has no such unit and the test is green in both arithmetics)
    hist = synth.synth_tensor("train.hist_t5b", (B, 1 + P, 7, 128), "normal")
    pred = pred.to(DEV)
    tp = TrainablePredictor(pred)
    assert tp.frozen_text and not any(n.startswith("predictor.text_encoder.") for n in tp.names)
    assert len(tp.all_names) > len(tp.names)
    tape = ag.Tape()
    preds = tp.rollout(tape, hist.to(DEV), ids.to(DEV), None, P, attn_masks=mask.to(DEV))
    stacked = ag.stack_frames(tape, preds)
    target = hist[:, 1:1 + P]
    total, sc = ag.mse(tape, stacked, target.to(DEV))
    tape.backward()
    # reference: the oracle's predictor step on the embeddings the frozen encoder produced here
    text = pred.encode_text_caption(caption_tokens=ids.to(DEV), attn_masks=mask.to(DEV)).detach().cpu().double()
    sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in pred.state_dict().items()
          if not k.startswith("predictor.text_encoder.") and v.dtype.is_floating_point}
    p = O.sub(sd, "predictor.")
    window, ref = hist[:, :1].double().clone(), []
    for t in range(P):
        cur = O.text_ocvp_step(p, window, text)
        window = torch.cat([window, cur.unsqueeze(1)], dim=1)
        ref.append(cur)
    ref_preds = torch.stack(ref, dim=1)
    ref_loss = F.mse_loss(ref_preds, target.double())
    ref_loss.backward()
    assert abs(total.item() * sc - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    assert rel_err(stacked.data, ref_preds) < 1e-4
    for name, var in tp.names.items():
        refg = sd[name].grad
        if refg is None:
            assert var.grad is None or var.grad.abs().max().item() == 0.0, name
            continue
        assert var.grad is not None and rel_err(var.grad, refg) < 2e-3, name


@pytest.mark.parametrize("name", ["VanillaTransformer", "OCVPSeq"])
def test_unconditioned_predictor_rollout_gradients_match_oracle_autograd(name):
    """ the unconditioned predictors the reference's trainer also accepts (OCVP.py: joint attention over all
    (frame, slot) tokens / object attention then time attention): BPTT gradients of every tensor against
    torch.autograd on the oracle (dropout off = the deterministic gradient), then three full optimisation steps """
    from oracle import slot_rollout_oracle as O
    from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
    from textocvp_amd.train import autograd as ag
    from textocvp_amd.train.predictor import TrainablePredictor
    from textocvp_amd.train.step import PredictorTrainStep
    Ks, P, B = 7, 3, 2
    exp = default_exp_params(num_slots=Ks, num_context=2, num_preds=P, predictor_name=name)
    pred = setup_predictor(exp)
    synth.fill_module_(pred, prefix=f"{name}.")
    hist = synth.synth_tensor("train.hist_u", (B, 2 + P, Ks, 128), "normal")
    sd = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point) for k, v in pred.state_dict().items()}
    ref_preds = O.rollout(sd, hist.double(), None, None, 2, P, buffer_size=pred.input_buffer_size, kind=name)
    target = hist[:, 2:2 + P]
    ref_loss = F.mse_loss(ref_preds, target.double())
    ref_loss.backward()
    pred = pred.to(DEV)
    tp = TrainablePredictor(pred, text_dropout=0.0)
    tape = ag.Tape()
    preds = tp.rollout(tape, hist.to(DEV), None, None, P)
    stacked = ag.stack_frames(tape, preds)
    total, sc = ag.mse(tape, stacked, target.to(DEV))
    tape.backward()
    assert abs(total.item() * sc - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    assert rel_err(stacked.data, ref_preds) < 1e-4
    for pname, var in tp.names.items():
        assert var.grad is not None and rel_err(var.grad, sd[pname].grad) < 2e-3, pname
    # full steps (frozen SAVi + image loss + clipped Adam), with the blocks' own dropout active
    savi = setup_model(exp["model"]).eval()
    synth.fill_module_(savi, prefix="savi.")
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    ts = PredictorTrainStep(savi.to(DEV), pred, lr=1e-4, clip=0.05, warmup_steps=0, generator=g)
    assert ts.model.text_dropout == pytest.approx(0.1)
    videos = synth.synth_videos(B, 2 + P, seed=0).to(DEV)
    noise = synth.synth_noise(B, Ks, 128, seed=1).to(DEV)
    losses = [ts.step(videos, None, None, init_noise=noise)["loss"] for _ in range(3)]
    assert all(v == v for v in losses) and losses[0] != losses[2]
    more = [ts.step_graphed(videos, None, None, init_noise=noise)["loss"] for _ in range(3)]    # capture + 2 replays
    assert all(v == v for v in more) and more[2] < losses[0]


def test_training_step_gradients_with_image_loss_match_oracle_autograd():
    """ the full loss of 04_train_predictor.py (image MSE through the frozen SAVi decoder + slot MSE):
    parameter gradients against torch.autograd on the CPU oracle, then one clipped Adam step
    against torch.optim.Adam on those reference gradients """
    from oracle import slot_rollout_oracle as O
    from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
    from textocvp_amd.train.step import PredictorTrainStep
    Ks, P, B = 7, 2, 2
    exp = default_exp_params(num_slots=Ks, num_context=1, num_preds=P)
    savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp)
    synth.fill_module_(savi, prefix="savi.")
    synth.fill_module_(pred, prefix="pred.")
    videos = synth.synth_videos(B, 1 + P, seed=0)
    tokens, lengths = synth.synth_captions(B, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(B, Ks, 128, seed=1)

    # reference: the CPU oracle (fp32, its SAVi half is not dtype-generic) differentiated by torch.autograd
    # on the predictor weights only (SAVi frozen)
    savi_sd = {k: v.detach() for k, v in savi.state_dict().items()}
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in pred.state_dict().items()}
    with torch.no_grad():
        hist = O.savi_decomp(savi_sd, videos, noise, 1 + P)
    preds = O.rollout(sd, hist, tokens, lengths, 1, P)
    imgs, _, _ = O.savi_decode(savi_sd, preds.reshape(B * P, Ks, 128), (64, 64), 3)
    l_img = F.mse_loss(imgs.view(B, P, 3, 64, 64), videos[:, 1:1 + P])
    l_slot = F.mse_loss(preds, hist[:, 1:1 + P])
    (l_img + l_slot).backward()

    savi, pred = savi.to(DEV), pred.to(DEV)
    ts = PredictorTrainStep(savi, pred, lr=1e-4, clip=0.05, warmup_steps=0, text_dropout=0.0)
    losses = ts.loss_and_grads(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))
    assert abs(losses["pred_img_mse"] - l_img.item()) < 2e-4 * abs(l_img.item())
    assert abs(losses["pred_slot_mse"] - l_slot.item()) < 2e-4 * abs(l_slot.item())
    worst, gref = 0.0, {}
    for name, var in ts.model.names.items():
        ref = sd[name].grad
        gref[name] = torch.zeros_like(sd[name]) if ref is None else ref
        if ref is None:
            assert var.grad is None or var.grad.abs().max().item() == 0.0, name
            continue
        e = rel_err(var.grad, ref)
        worst = max(worst, e)
        assert e < 5e-3, (name, e)
    print(f"training step (image + slot loss): worst relative gradient error {worst:.2e}; "
          f"losses img {losses['pred_img_mse']:.5f} slot {losses['pred_slot_mse']:.5f}")

    # optimiser: clip_grad_norm_(0.05) + Adam against torch.optim.Adam fed with the SAME gradients (the first
    # Adam step is ~lr * sign(g): elements with |g| ~ eps would otherwise amplify the 1e-6 gradient noise)
    before = {n: v.data.detach().cpu().double().clone() for n, v in ts.model.names.items()}
    mine = {n: (torch.zeros_like(before[n]) if v.grad is None else v.grad.detach().cpu().double().clone())
            for n, v in ts.model.names.items()}
    norm, lr = ts.apply()
    ref_params = [torch.nn.Parameter(before[n].clone()) for n in ts.model.names]
    for p_, n in zip(ref_params, ts.model.names):
        p_.grad = mine[n].clone()
    ref_norm = torch.nn.utils.clip_grad_norm_(ref_params, 0.05)
    torch.optim.Adam(ref_params, lr=1e-4).step()
    assert abs(norm - ref_norm.item()) < 1e-4 * ref_norm.item()
    for p_, (n, v) in zip(ref_params, ts.model.names.items()):
        upd_ref = (p_.detach() - before[n])
        upd = v.data.detach().cpu().double() - before[n]
        assert (upd - upd_ref).abs().max().item() < 2e-7, n                     # |update| <= lr = 1e-4
    # the module really holds the updated weights and derived caches follow them
    name0 = next(iter(ts.model.names))
    assert torch.equal(dict(pred.named_parameters())[name0].data, ts.model.names[name0].data)
    ts.loss_and_grads(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))


def _build_step(Ks=7, P=2):
    from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
    from textocvp_amd.train.step import PredictorTrainStep
    exp = default_exp_params(num_slots=Ks, num_context=1, num_preds=P)
    savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp)
    synth.fill_module_(savi, prefix="savi.")
    synth.fill_module_(pred, prefix="pred.")
    ts = PredictorTrainStep(savi.to(DEV), pred.to(DEV), lr=1e-4, clip=0.05, warmup_steps=0, text_dropout=0.0)
    videos = synth.synth_videos(2, 1 + P, seed=0)
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(2, Ks, 128, seed=1)
    return ts, videos, tokens, lengths, noise


def test_textocvp_t5_training_step_eager_and_graphed():
    """ full optimisation steps of TextOCVP_T5 (frozen T5 encoder, `attn_masks` as in predictor_wrapper.py:101-111):
    eager and graph-replayed steps agree, only the predictor's own tensors move and carry Adam moments, the
    optimiser state keeps torch.optim's indices over ALL parameters """
    from conftest import load_golden
    from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
    from textocvp_amd.train.step import PredictorTrainStep
    g = load_golden("t5_encoder.npz")
    ids, mask = torch.from_numpy(g["ids"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    B, P, Ks = ids.shape[0], 2, 7
    res = []
    for graphed in (False, True):
        exp = default_exp_params(num_slots=Ks, num_context=1, num_preds=P, predictor_name="TextOCVP_T5")
        savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp)
        synth.fill_module_(savi, prefix="savi.")
        synth.fill_module_(pred.predictor.text_encoder, prefix="t5.")
        for part in ("predictor", "mlp_in", "mlp_out", "pe"):      # everything outside the T5 encoder, incl. the learned PE
            synth.fill_module_(getattr(pred.predictor, part), prefix=f"pred.predictor.{part}.")
        ts = PredictorTrainStep(savi.to(DEV), pred.to(DEV), lr=1e-4, clip=0.05, warmup_steps=0)
        t5_before = {n: p.detach().clone() for n, p in pred.named_parameters() if n.startswith("predictor.text_encoder.")}
        videos = synth.synth_videos(B, 1 + P, seed=0).to(DEV)
        noise = synth.synth_noise(B, Ks, 128, seed=1).to(DEV)
        run = ts.step_graphed if graphed else ts.step
        losses = [dict(run(videos, ids, None, attn_masks=mask, init_noise=noise)) for _ in range(3)]
        assert all(torch.equal(p, t5_before[n]) for n, p in pred.named_parameters() if n in t5_before)
        assert set(ts.state) == set(ts.model.names) and not any(n.startswith("predictor.text_encoder.") for n in ts.state)
        opt = ts.optimizer_state_dict()
        assert len(opt["param_groups"][0]["params"]) == len(ts.model.all_names) > len(opt["state"])
        res.append(losses)
    for a, b in zip(*res):
        assert abs(a["loss"] - b["loss"]) < 1e-5 * abs(a["loss"]) and a["lr"] == b["lr"]
    assert res[0][0]["loss"] != res[0][2]["loss"]


def test_training_step_against_reference_golden():
    """ losses and gradients of the reference's own training step (torch.autograd on the reference
    modules, tests/golden/train_c5.npz) """
    from conftest import load_golden
    g = load_golden("train_c5.npz")
    ts, videos, tokens, lengths, noise = _build_step()
    losses = ts.loss_and_grads(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))
    assert abs(losses["pred_img_mse"] - float(g["loss_img"])) < 2e-4 * float(g["loss_img"])
    assert abs(losses["pred_slot_mse"] - float(g["loss_slot"])) < 2e-4 * float(g["loss_slot"])
    worst = 0.0
    for name, ref_norm in zip(g["names"], g["grad_norms"]):
        v = ts.model.names[str(name)]
        norm = 0.0 if v.grad is None else float(v.grad.norm())
        e = abs(norm - float(ref_norm)) / max(float(ref_norm), 1e-8)
        worst = max(worst, e if float(ref_norm) > 1e-7 else 0.0)
        assert e < 5e-3 or abs(norm - float(ref_norm)) < 1e-8, (str(name), norm, float(ref_norm))
    for key in g:
        if key.startswith("grad::"):
            grad = ts.model.names[key[6:]].grad
            if grad.dim() == 2 and grad.numel() > 40000:
                grad = grad[::4, ::4]
            ref = torch.from_numpy(g[key])
            assert rel_err(grad.reshape(ref.shape), ref) < 5e-3, key
    print(f"vs reference golden: worst gradient-norm error {worst:.2e}")


def _ddp_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ts, videos, tokens, lengths, noise = _build_step()
    sl = slice(rank, rank + 1)                                  # one sequence per rank, same padded captions
    ts.loss_and_grads(videos[sl].to(DEV), tokens[sl].to(DEV), lengths[sl].to(DEV), init_noise=noise[sl].to(DEV))
    ts.all_reduce_grads()
    if rank == 0:
        torch.save({n: v.grad.cpu() for n, v in ts.model.names.items() if v.grad is not None},
                   os.path.join(out_dir, "avg.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch(tmp_path):
    """ data-parallel step: two ranks (gloo, both on this GPU) with one sequence each + the flat
    all-reduce == the gradient of the two-sequence batch (MSE means over equal shares) """
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    avg = torch.load(tmp_path / "avg.pt")
    ts, videos, tokens, lengths, noise = _build_step()
    ts.loss_and_grads(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))
    for name, v in ts.model.names.items():
        if v.grad is None:
            continue
        assert rel_err(avg[name], v.grad.cpu()) < 2e-4, name


def test_dropout_and_attention_probability_dropout():
    """ nn.Dropout / nn.MultiheadAttention(dropout=p) in training mode with a shared uniform sample """
    ag = _ag()
    p = 0.1
    x, gy = rnd("dx", (5, 12, 128)), rnd("dgy", (5, 12, 128))
    r = torch.rand(5, 12, 128, generator=torch.Generator().manual_seed(3))
    xr = x.double().requires_grad_()
    y = xr * (r >= p).double() / (1 - p)
    y.backward(gy.double())
    tape = ag.Tape()
    X = ag.Var(x.to(DEV), True)
    Y = ag.dropout(tape, X, p, sample=r.to(DEV))
    Y.grad = gy.to(DEV)
    tape.backward()
    assert rel_err(Y.data, y) < 1e-6 and rel_err(X.grad, xr.grad) < 1e-6
    kept = (Y.data != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.02
    # attention with dropout on the probabilities, key padding
    B, H, T, E = 3, 4, 12, 128
    q, k, v, go = rnd("uq", (B, T, E)), rnd("uk", (B, T, E)), rnd("uv", (B, T, E)), rnd("ugo", (B, T, E))
    lens = [12, 5, 9]
    rs = torch.rand(B, H, T, T, generator=torch.Generator().manual_seed(4))
    scale = (E // H) ** -0.5
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))

    def heads(t):
        return t.reshape(B, T, H, E // H).transpose(1, 2)
    sc = heads(qr) @ heads(kr).transpose(-1, -2) * scale
    mask = torch.arange(T)[None, :] >= torch.tensor(lens)[:, None]
    pr = torch.softmax(sc.masked_fill(mask[:, None, None, :], float("-inf")), -1)
    o = ((pr * (rs >= p).double() / (1 - p)) @ heads(vr)).transpose(1, 2).reshape(B, T, E)
    o.backward(go.double())
    tape = ag.Tape()
    Q, Kv, V = (ag.Var(t.to(DEV), True) for t in (q, k, v))
    O = ag.attention_unfused(tape, Q, Kv, V, H, scale, key_len=torch.tensor(lens, dtype=torch.int32, device=DEV),
                             p_drop=p, sample=rs.to(DEV))
    O.grad = go.to(DEV)
    tape.backward()
    assert rel_err(O.data, o) < 1e-5
    for got, ref in ((Q, qr), (Kv, kr), (V, vr)):
        assert rel_err(got.grad, ref.grad) < 2e-5


def test_training_step_with_text_dropout_is_reproducible_and_close():
    """ default step = the reference's training mode (caption-encoder dropout 0.1): same generator seed ->
    same gradients; the loss stays near the deterministic one """
    from textocvp_amd.train.step import PredictorTrainStep
    ts0, videos, tokens, lengths, noise = _build_step()
    base = ts0.loss_and_grads(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))
    outs = []
    for _ in range(2):
        g = torch.Generator(device=DEV).manual_seed(11)
        ts = PredictorTrainStep(ts0.savi, ts0.wrapper, lr=1e-4, warmup_steps=0, generator=g)
        assert ts.model.text_dropout == pytest.approx(0.1)
        losses = ts.loss_and_grads(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))
        outs.append((losses, {n: v.grad.clone() for n, v in ts.model.names.items() if v.grad is not None}))
    assert outs[0][0] == outs[1][0]
    for n in outs[0][1]:      # same dropout masks; the LDS / global float atomics of two kernels reorder sums
        assert rel_err(outs[0][1][n], outs[1][1][n].cpu()) < 1e-5, n
    assert abs(outs[0][0]["loss"] - base["loss"]) < 0.05 * base["loss"]
    assert outs[0][0]["loss"] != base["loss"]


@pytest.mark.parametrize("buffer,teacher", [(2, False), (10, True)])
def test_rollout_gradients_sliding_window_and_teacher_forcing(buffer, teacher):
    """ window that slides (buffer 2 < 1 + 4 preds) and teacher forcing (ground-truth slots re-enter the
    window, no gradient through them): gradients against torch.autograd on the oracle """
    from oracle import slot_rollout_oracle as O
    from textocvp_amd.setup_model import default_exp_params, setup_predictor
    from textocvp_amd.train import autograd as ag
    from textocvp_amd.train.predictor import TrainablePredictor
    P, Ks, B = 4, 7, 2
    exp = default_exp_params(num_slots=Ks, num_context=1, num_preds=P, input_buffer_size=buffer,
                             teacher_force=teacher)
    pred = setup_predictor(exp)
    synth.fill_module_(pred, prefix="pred.")
    hist = synth.synth_tensor("train.hist2", (B, 1 + P, Ks, 128), "normal")
    tokens, lengths = synth.synth_captions(B, max_len=12, lengths=[9, 12], seed=0)
    sd = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in pred.state_dict().items()}
    ref_preds = O.rollout(sd, hist.double(), tokens, lengths, 1, P, buffer_size=buffer, teacher_force=teacher)
    F.mse_loss(ref_preds, hist[:, 1:1 + P].double()).backward()
    pred = pred.to(DEV)
    tp = TrainablePredictor(pred, text_dropout=0.0)
    tape = ag.Tape()
    stacked = ag.stack_frames(tape, tp.rollout(tape, hist.to(DEV), tokens.to(DEV), lengths.to(DEV), P))
    ag.mse(tape, stacked, hist[:, 1:1 + P].to(DEV))
    tape.backward()
    assert rel_err(stacked.data, ref_preds) < 1e-4
    for name, var in tp.names.items():
        ref = sd[name].grad
        if ref is None or ref.abs().max().item() == 0.0:
            continue
        # A pre-activation within ~1e-6 of zero can land on the other side of the ReLU in the fp64
        # reference (about 1e-6 of the 4.6 M hidden activations here); with only ~70 tokens one such
        # flip moves a whole row of a weight gradient by percents.  Demand element-wise agreement
        # everywhere except on a vanishing fraction of elements.
        err = (var.grad.detach().cpu().double() - ref).abs() / ref.abs().max()
        assert (err > 5e-3).double().mean().item() < 2e-3, (name, err.max().item())
        assert err.max().item() < 0.2, (name, err.max().item())


def test_graph_replayed_steps_equal_eager_steps():
    """ three optimisation steps issued launch by launch == one eager step + two replays of the captured
    HIP graphs (same batches, dropout off): same losses, same weights """
    res = []
    for graphed in (False, True):
        ts, videos, tokens, lengths, noise = _build_step()
        args = (videos.to(DEV), tokens.to(DEV), lengths.to(DEV))
        run = ts.step_graphed if graphed else ts.step
        losses = [run(*args, init_noise=noise.to(DEV)) for _ in range(3)]
        res.append((losses, {n: v.data.detach().cpu().clone() for n, v in ts.model.names.items()}))
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a["loss"] - b["loss"]) < 1e-5 * abs(a["loss"])
        assert abs(a["grad_norm"] - b["grad_norm"]) < 1e-4 * a["grad_norm"]
        assert a["lr"] == b["lr"]
    assert res[0][0][0]["loss"] != res[0][0][2]["loss"]                       # the weights really moved
    for n in res[0][1]:
        # three steps of at most lr = 1e-4.  Not bit-identical: two float-atomic reductions in the backward pass make
        # gradients reproducible to ~1e-10, and two things amplify that: Adam turns a near-zero gradient g into
        # lr * g / (|g| + eps) (lr / eps = 1e4, up to a flipped sign of a whole update), and a hidden unit whose
        # pre-activation sits within that noise of zero is switched on in one run and off in the other -- then its whole
        # weight row (512 elements) moves by +-lr per step.  scripts/train_determinism_probe.py shows the same two
        # outcomes between two EAGER runs (and between two replayed ones): it is the step, not the replay.  So: every
        # element within the 3-step worst case, all but 0.1 % of a tensor within 1e-5
        d = (res[0][1][n] - res[1][1][n]).abs()
        assert d.max().item() <= 6.1e-4, n
        assert int((d > 1e-5).sum()) <= max(2, d.numel() // 1000), (n, int((d > 1e-5).sum()))


@pytest.mark.parametrize("M,N,Kd,tA,tB,ld_pad", [(70, 50, 33, False, False, 0), (64, 64, 64, True, False, 0),
                                               (130, 36, 300, False, True, 0), (37, 128, 20, True, True, 4),
                                               (300, 300, 64, False, True, 0), (45, 31, 18, True, False, 1)])
def test_bmm_generic(M, N, Kd, tA, tB, ld_pad):
    """ strided / batched fp32 GEMM: both transpose flags, ragged edges, aligned (16-byte loads) and odd leading
    dimensions (element loads), accumulate, alpha """
    ag = _ag()
    nb1, nb2 = 2, 3
    a_shape = (Kd, M + ld_pad) if tA else (M, Kd + ld_pad)
    b_shape = (N, Kd + ld_pad) if tB else (Kd, N + ld_pad)
    A = rnd("bA", (nb1, nb2) + a_shape)
    B = rnd("bB", (nb1, nb2) + b_shape)
    C0 = rnd("bC", (nb1, nb2, M, N))
    Aop = A[..., :M].transpose(-1, -2) if tA else A[..., :Kd]
    Bop = B[..., :Kd].transpose(-1, -2) if tB else B[..., :N]
    ref = 0.5 * (Aop.double() @ Bop.double()) + C0.double()
    Ad, Bd, Cd = A.to(DEV), B.to(DEV), C0.to(DEV).clone()
    ag.bmm(Ad, Bd, Cd, M, N, Kd, a_shape[1], b_shape[1], N, transA=tA, transB=tB, batch=(nb1, nb2),
           sA=(nb2 * a_shape[0] * a_shape[1], a_shape[0] * a_shape[1]),
           sB=(nb2 * b_shape[0] * b_shape[1], b_shape[0] * b_shape[1]), sC=(nb2 * M * N, M * N), alpha=0.5, acc=True)
    assert rel_err(Cd, ref) < 1e-5


def test_training_first_pass_is_range_checked_and_falls_back():
    """ weights / activations outside the fp16-plane range (|x| < 255) must not be trained on silently: the first
    pass is range-checked and the tripping arithmetic moves to its fp32-range fallback (here: a predictor whose
    input projection is scaled so that the tokens exceed 255) """
    import warnings
    ts, videos, tokens, lengths, noise = _build_step()
    with torch.no_grad():
        ts.wrapper.predictor.mlp_in.weight.mul_(400.0)
    ts.model.mark_updated()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = ts.step(videos.to(DEV), tokens.to(DEV), lengths.to(DEV), init_noise=noise.to(DEV))
    assert ts.model.precision == "bf16x6" and any("fp32-range fallback" in str(x.message) for x in w)
    assert out["loss"] == out["loss"] and ts._range_ok                      # finite, and checked only once
