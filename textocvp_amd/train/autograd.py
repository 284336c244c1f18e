"""
Minimal tape autograd over the HIP kernels, for the predictor training step (SURVEY.md section 8f
rank 2; reference 04_train_predictor.py:57-108).  torch.autograd is not used: every forward op
calls a libtocvp kernel and records a closure that calls the matching backward kernels; gradient
accumulation is an axpby kernel.  torch provides device memory (empty / clone / index copies) only.

    tape = Tape()
    y = linear(tape, x, W, b, act=ACT_RELU)
    ...
    loss = mse(tape, pred, target)          # python float + seeds d(loss)/d(pred)
    tape.backward()                         # walks the closures in reverse
"""

import os

import torch

from .. import kernels as K

__all__ = ["Var", "Tape", "linear", "layer_norm", "attention", "attention_unfused", "dropout", "add",
           "activation", "add_position_rows", "stack_frames", "take_frame", "take_last_tokens", "transpose12",
           "reshape", "embedding",
           "mask_rows", "mse",
           "accumulate", "bmm"]

_L = K.lib
_INDEX_CACHE = {}
# weight / bias gradients of the large linears on the transpose-free split-K kernel (tocvp_gemm_tn_f32)
_TN = os.environ.get("TOCVP_TRAIN_TN", "1") != "0"
_TN_TARGET_WGS = 512          # two workgroups per CU
_FUSED_ATTN_BWD = os.environ.get("TOCVP_TRAIN_FUSED_ATTN_BWD", "1") != "0"
# arithmetic of the data-gradient GEMMs dx = g W: "bf16x6" = three bf16 planes per operand, six products (fp32-class,
# ~2^-24 per product); "bf16x3" = two planes, three products (~2^-17 per product: a 16-bit mantissa, finer than the
# TF32 convolutions / matmuls a stock PyTorch training run of the reference uses on its GPUs).  No range to respect
# either way (bf16 planes keep the fp32 exponent), so gradients need no scale.  Default since round 3: three products --
# the golden gradient norms (253 tensors at 2e-4) and losses (1e-5) of tests/golden/train_c5.npz hold, the step goes
# from 581 to 554 ms at the configs[1] shapes (data-gradient GEMMs 85 -> ~55 ms).  TOCVP_TRAIN_DGRAD=bf16x6 restores six.
_DGRAD_PRECISION = os.environ.get("TOCVP_TRAIN_DGRAD", "bf16x3")
# weight gradients dW = g^T x of the large linears: "bf16x3" = the split-operand kernel (tocvp_gemm_tn_bf16x3_f32, the
# arithmetic of the data gradients above) where the row count allows (M % 32 == 0), "fp32" = the exact fp32 MFMA
_WGRAD_PRECISION = os.environ.get("TOCVP_TRAIN_WGRAD", "bf16x3")


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


class Var:
    """
    a tensor on the tape: ``data`` (contiguous fp32 CUDA), ``grad`` (None until something flows in).
    An op may attach ``recompute`` (a closure that rebuilds ``data`` bit for bit from the op's inputs); the
    model code can then ``release()`` the activation once its forward consumers have run -- the first
    backward closure that reads ``data`` rebuilds it, and the producing op drops it again when its own
    backward has run (it runs after every consumer's).  Activation memory traded for one cheap kernel.
    """
    __slots__ = ("_data", "grad", "requires_grad", "name", "recompute", "single_use", "gated")

    def __init__(self, data, requires_grad=False, name=None):
        assert data.is_cuda and data.dtype == torch.float32
        self._data = data if data.is_contiguous() else data.contiguous()
        self.grad, self.requires_grad, self.name, self.recompute = None, requires_grad, name, None
        # single_use: set by the model code on a ReLU output that feeds exactly one linear -- that linear's
        # backward may then write the gradient already masked by the ReLU (gated), saving the masking pass
        self.single_use, self.gated = False, False

    @property
    def data(self):
        if self._data is None:
            self._data = self.recompute()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def release(self):
        if self.recompute is not None:
            self._data = None

    @property
    def shape(self):
        return self.data.shape


class Tape:
    def __init__(self):
        self.nodes = []
        self.cache = {}          # per-step derived tensors (transposed weights of the backward GEMMs)

    def record(self, fn):
        self.nodes.append(fn)

    def backward(self):
        for fn in reversed(self.nodes):
            fn()
        for key, ent in self.cache.items():
            if isinstance(key, tuple) and key[0] == "tn":
                _finish_weight_grad(ent)
            elif isinstance(key, tuple) and key[0] == "ln":
                if ent["gamma"].requires_grad:
                    accumulate(ent["gamma"], colsum(ent["pg"]))
                if ent["beta"].requires_grad:
                    accumulate(ent["beta"], colsum(ent["pb"]))
        self.nodes = []
        self.cache = {}


# ------------------------------------------------------------------------------------------------
# raw kernel wrappers
# ------------------------------------------------------------------------------------------------
def axpby(x, y, a=1.0, b=1.0):
    """ y = a * x + b * y (in place on y) """
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    K._check(_L().tocvp_axpby_f32(_p(x), _p(y), x.numel(), float(a), float(b), _s()), "tocvp_axpby_f32")
    return y


def accumulate(var, g):
    """ add gradient ``g`` (same shape as var.data) into var.grad; takes ownership of ``g`` when it is the first """
    if not var.requires_grad:
        return
    g = g.reshape(var.data.shape)
    if var.grad is None:
        var.grad = g if g.is_contiguous() else g.contiguous()
    else:
        axpby(g if g.is_contiguous() else g.contiguous(), var.grad, 1.0, 1.0)


def bmm(A, B, C, M, N, Kd, lda, ldb, ldc, transA=False, transB=False, batch=(1, 1),
        sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, acc=False):
    K._check(_L().tocvp_bmm_f32(_p(A), lda, sA[0], sA[1], int(transA), _p(B), ldb, sB[0], sB[1], int(transB),
                                _p(C), ldc, sC[0], sC[1], batch[0], batch[1], M, N, Kd, float(alpha),
                                int(acc), _s()), "tocvp_bmm_f32")
    return C


def colsum(x2, out=None, acc=False):
    """ x2 (rows, cols) contiguous -> (cols,) column sums, deterministic two-stage """
    rows, cols = x2.shape
    chunk = 32 if rows >= 2048 else max(1, (rows + 63) // 64)      # enough workgroups to fill the chip
    nch = (rows + chunk - 1) // chunk
    part = torch.empty((nch, cols), device=x2.device, dtype=torch.float32)
    K._check(_L().tocvp_colsum_partial_f32(_p(x2), _p(part), rows, cols, cols, chunk, _s()),
             "tocvp_colsum_partial_f32")
    res = torch.empty((1, cols), device=x2.device, dtype=torch.float32)
    K._check(_L().tocvp_colsum_partial_f32(_p(part), _p(res), nch, cols, cols, nch, _s()),
             "tocvp_colsum_partial_f32")
    res = res.reshape(cols)
    if out is None:
        return res
    return axpby(res, out, 1.0, 1.0 if acc else 0.0)


def _sum_splits(part):
    """ (splits, n) -> (n,): the splits added in index order by one launch """
    splits, n = part.shape
    res = torch.empty((1, n), device=part.device, dtype=torch.float32)
    K._check(_L().tocvp_colsum_partial_f32(_p(part), _p(res), splits, n, n, splits, _s()),
             "tocvp_colsum_partial_f32")
    return res.reshape(n)


# Weight gradients of ONE weight over all rollout steps in ONE launch (round 5): back-propagation through time uses every linear
# once per step (04_train_predictor.py:57-108: predictions are fed back un-detached), and dW = sum_t g_t^T x_t is one product
# over the concatenated rows -- tocvp_gemm_tn_bf16x3_multi_f32 takes up to 20 (g_t, x_t) segments as kernel arguments.  The
# backward closures only PARK their (g, x) pair (both stay alive until the launch); a weight is flushed when 20 pairs wait
# and when the tape finishes.  1583 launches of ~45 us per step became ~100 of long reductions.  TOCVP_TRAIN_WGRAD_DEFER=0:
# one launch per use.
_WGRAD_DEFER = os.environ.get("TOCVP_TRAIN_WGRAD_DEFER", "1") != "0"
_WGRAD_MAXSEG = 20


def _flush_weight_grad(ent):
    pend = ent.pop("pending", None)
    if not pend:
        return
    import ctypes
    n = len(pend)
    N, Kd = pend[0][0].shape[1], pend[0][1].shape[1]
    G = (ctypes.c_void_p * n)(*[g.data_ptr() for g, _ in pend])
    X = (ctypes.c_void_p * n)(*[x.data_ptr() for _, x in pend])
    rows = (ctypes.c_int * n)(*[g.shape[0] for g, _ in pend])
    total = sum(g.shape[0] for g, _ in pend)
    K._timed("gemm_tn", 2.0 * total * N * Kd, lambda: K._check(
        _L().tocvp_gemm_tn_bf16x3_multi_f32(ctypes.cast(G, ctypes.c_void_p), ctypes.cast(X, ctypes.c_void_p),
                                            ctypes.cast(rows, ctypes.c_void_p), n, N, Kd, _p(ent["part"]), _p(ent["bias"]), N, Kd,
                                            ent["splits"], 1 if ent["written"] else 0, _s()),
        "tocvp_gemm_tn_bf16x3_multi_f32"))
    ent["written"] = True


def _weight_grad_tn(tape, W, b, g, x2):
    """
    dW += g^T x2 (and db += column sums of g) on tocvp_gemm_tn_f32.  The split-K partial sums of one weight
    live in ONE buffer per backward pass (tape.cache): every rollout step accumulates into it and
    Tape.backward() adds the splits into W.grad / b.grad once at the end.
    """
    M, N = g.shape
    Kd = x2.shape[1]
    want_b = b is not None and b.requires_grad
    ent = tape.cache.get(("tn", id(W)))
    first = ent is None
    if first:
        tiles = (N // 128) * (Kd // 128)
        splits = max(1, min(16, -(-_TN_TARGET_WGS // tiles), M // 64))
        ent = tape.cache[("tn", id(W))] = {
            "W": W, "b": b if want_b else None, "splits": splits, "written": False,
            "part": torch.empty((splits, N * Kd), device=g.device, dtype=torch.float32),
            "bias": torch.empty((splits, N), device=g.device, dtype=torch.float32) if want_b else None}
    if _WGRAD_DEFER and _WGRAD_PRECISION == "bf16x3" and M % 32 == 0 and g.is_contiguous() and x2.is_contiguous():
        ent.setdefault("pending", []).append((g, x2))
        if len(ent["pending"]) >= _WGRAD_MAXSEG:
            _flush_weight_grad(ent)
            return False
        return True                                       # g and x2 are parked: nobody may write into them any more
    # the first write covers every slice; later (accumulating) uses with few rows touch only as many slices as they can keep busy
    fresh = not ent["written"]
    active = ent["splits"] if fresh else max(1, min(ent["splits"], M // 128))
    if _WGRAD_PRECISION == "bf16x3" and M % 32 == 0:
        K._timed("gemm_tn", 2.0 * M * N * Kd, lambda: K._check(
            _L().tocvp_gemm_tn_bf16x3_f32(_p(g), N, _p(x2), Kd, _p(ent["part"]), _p(ent["bias"]), M, N, Kd,
                                          active, 0 if fresh else 1, _s()), "tocvp_gemm_tn_bf16x3_f32"))
    else:
        K._timed("gemm_tn", 2.0 * M * N * Kd, lambda: K._check(
            _L().tocvp_gemm_tn_f32(_p(g), N, _p(x2), Kd, _p(ent["part"]), _p(ent["bias"]), M, N, Kd,
                                   active, 0 if fresh else 1, _s()), "tocvp_gemm_tn_f32"))
    ent["written"] = True
    return False


def _finish_weight_grad(ent):
    _flush_weight_grad(ent)
    W, b = ent["W"], ent["b"]
    dW = _sum_splits(ent["part"]).reshape(W.data.shape)
    if W.grad is None:
        W.grad = dW
    else:
        axpby(dW, W.grad, 1.0, 1.0)
    if b is not None:
        db = _sum_splits(ent["bias"])
        if b.grad is None:
            b.grad = db
        else:
            axpby(db, b.grad, 1.0, 1.0)


# ------------------------------------------------------------------------------------------------
# differentiable ops
# ------------------------------------------------------------------------------------------------
def linear(tape, x, W, b=None, act=K.ACT_NONE, precision="f16x3", residual=None):
    """
    y = act(x W^T + b) (+ residual); x (..., K), W (N, K).  ReLU is fused; GELU keeps the pre-activation.
    residual: a Var of the output's shape added in the GEMM epilogue (no activation then); its gradient is
    the output gradient itself.
    """
    N, Kd = W.data.shape
    fused = act if act == K.ACT_RELU else K.ACT_NONE
    assert residual is None or act == K.ACT_NONE
    pre = K.linear(x.data, W.data, None if b is None else b.data, act=fused, precision=precision,
                   residual=None if residual is None else residual.data)
    y = pre
    if act == K.ACT_GELU:
        y = torch.empty_like(pre)
        K._check(_L().tocvp_act_f32(_p(pre), _p(y), pre.numel(), K.ACT_GELU, _s()), "tocvp_act_f32")
    out = Var(y, x.requires_grad or W.requires_grad or (b is not None and b.requires_grad)
              or (residual is not None and residual.requires_grad))
    if not out.requires_grad:
        return out
    if act != K.ACT_GELU:
        out.recompute = lambda: K.linear(x.data, W.data, None if b is None else b.data, act=fused,
                                         precision=precision,
                                         residual=None if residual is None else residual.data)
        pre = y = None                                   # the ReLU mask is read from out.data

    def backward():
        if out.grad is None:
            out.release()
            return
        g = out.grad.reshape(-1, N)
        if act != K.ACT_NONE and not (act == K.ACT_RELU and out.gated):
            gg = torch.empty_like(g)
            K._check(_L().tocvp_act_bwd_f32(_p(g), _p(out.data if pre is None else pre), _p(gg), g.numel(),
                                            int(act), _s()), "tocvp_act_bwd_f32")
            g = gg
        out.release()                                     # every consumer's backward has run
        M = g.shape[0]
        x2 = x.data.reshape(M, Kd)
        # Large, aligned shapes run on the split-operand GEMM of the forward pass (bf16x6: fp32-class and,
        # unlike the fp16 planes, with the fp32 exponent range that small gradients need); the operands
        # it wants transposed are copied (data movement).  Everything else takes the generic fp32 kernel.
        fast = M >= 256 and M % 64 == 0 and N % 64 == 0 and Kd % 64 == 0
        tn = _TN and W.requires_grad and M >= 256 and M % 16 == 0 and N % 128 == 0 and Kd % 128 == 0
        parked = False
        if tn:                                            # dW and db in one transpose-free launch (or parked for it)
            parked = _weight_grad_tn(tape, W, b, g, x2)
        elif W.requires_grad:                             # dW (N, K) = g^T (N, M) x (M, K)
            splits = min(16, M // 512)
            if fast and M >= 8192 and (N // 64) * (Kd // 64) < 128 and M % splits == 0:
                # small weight, long reduction: a single GEMM has too few output tiles to fill the chip.
                # Split-K: one batched launch writes `splits` partial products, a column sum adds them.
                Mc = M // splits
                part = torch.empty((splits, N * Kd), device=g.device, dtype=torch.float32)
                bmm(g, x2, part, N, Kd, Mc, N, Kd, Kd, transA=True, batch=(splits, 1),
                    sA=(Mc * N, 0), sB=(Mc * Kd, 0), sC=(N * Kd, 0))
                dW = colsum(part).reshape(N, Kd)
                if W.grad is None:
                    W.grad = dW
                else:
                    axpby(dW, W.grad, 1.0, 1.0)
            elif fast:
                gT, xT = g.t().contiguous(), x2.t().contiguous()
                if W.grad is None:
                    W.grad = K.linear(gT, xT, precision="bf16x6")
                else:
                    K.linear(gT, xT, residual=W.grad, out=W.grad, precision="bf16x6")
            elif W.grad is None:
                W.grad = torch.empty_like(W.data)
                bmm(g, x2, W.grad, N, Kd, M, N, Kd, Kd, transA=True)
            else:
                bmm(g, x2, W.grad, N, Kd, M, N, Kd, Kd, transA=True, acc=True)
        if b is not None and b.requires_grad and not tn:
            if b.grad is None:
                b.grad = colsum(g)
            else:
                colsum(g, out=b.grad, acc=True)
        if x.requires_grad:                               # dx (M, K) = g (M, N) W (N, K)
            if fast:
                Wt = tape.cache.get(id(W))          # the same weight is used by every rollout step
                if Wt is None:
                    Wt = tape.cache[id(W)] = W.data.t().contiguous()
                if x.grad is not None and x.grad.is_contiguous():      # add into the gradient in the epilogue
                    xg = x.grad.reshape(M, Kd)
                    K.linear(g, Wt, residual=xg, out=xg, precision=_DGRAD_PRECISION)
                elif x.grad is None and x.single_use and Kd % 32 == 0 and N % 64 == 0:
                    # x is a ReLU output with no other consumer: the gradient leaves the GEMM already masked
                    x.grad = K.linear(g, Wt, residual=x2, act=K.ACT_GATE, precision=_DGRAD_PRECISION).reshape(x.data.shape)
                    x.gated = True
                else:
                    accumulate(x, K.linear(g, Wt, precision=_DGRAD_PRECISION))
            else:
                dx = torch.empty((M, Kd), device=g.device, dtype=torch.float32)
                bmm(g, W.data, dx, M, Kd, N, N, Kd, Kd)
                accumulate(x, dx)
        if residual is not None:                          # last: the residual branch may take over out.grad
            # (a parked gradient stays this weight's operand until its launch: the residual branch, whose later consumers add
            # into what it owns IN PLACE, gets a copy)
            accumulate(residual, out.grad.clone() if parked and g.data_ptr() == out.grad.data_ptr() else out.grad)
    tape.record(backward)
    return out


def activation(tape, x, act):
    y = torch.empty_like(x.data)
    K._check(_L().tocvp_act_f32(_p(x.data), _p(y), y.numel(), int(act), _s()), "tocvp_act_f32")
    out = Var(y, x.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            dx = torch.empty_like(x.data)
            K._check(_L().tocvp_act_bwd_f32(_p(out.grad), _p(x.data), _p(dx), dx.numel(), int(act), _s()),
                     "tocvp_act_bwd_f32")
            accumulate(x, dx)
        tape.record(backward)
    return out


def layer_norm(tape, x, gamma, beta, eps):
    out = Var(K.layer_norm(x.data, gamma.data, beta.data, eps),
              x.requires_grad or gamma.requires_grad or beta.requires_grad)
    if not out.requires_grad:
        return out
    out.recompute = lambda: K.layer_norm(x.data, gamma.data, beta.data, eps)
    D = x.data.shape[-1]

    def backward():
        out.release()                                     # every consumer's backward has run
        if out.grad is None:
            return
        rows = x.data.numel() // D
        nwaves = min(1024, (rows + 3) // 4 * 4)
        nwaves = max(4, nwaves // 4 * 4)
        into = x.requires_grad and x.grad is not None and x.grad.is_contiguous()     # add into it in the kernel
        dx = x.grad if into else torch.empty_like(x.data)
        # per-wave partial sums of dgamma / dbeta: one zero-initialised buffer per parameter pair and backward
        # pass, every use of the LayerNorm adds into it, Tape.backward() column-sums it once
        ent = tape.cache.get(("ln", id(gamma)))
        if ent is None:
            ent = tape.cache[("ln", id(gamma))] = {
                "gamma": gamma, "beta": beta,
                "pg": torch.zeros((1024, D), device=dx.device, dtype=torch.float32),
                "pb": torch.zeros((1024, D), device=dx.device, dtype=torch.float32)}
        K._check(_L().tocvp_layernorm_bwd_f32(_p(x.data), _p(gamma.data), _p(out.grad), _p(dx), _p(ent["pg"]),
                                              _p(ent["pb"]), nwaves, rows, D, float(eps), 3 if into else 1, _s()),
                 "tocvp_layernorm_bwd_f32")
        if not into:
            accumulate(x, dx)
    tape.record(backward)
    return out


def attention(tape, q, k, v, heads, scale, key_len=None):
    """ multi-head softmax attention: q (B, Tq, E), k / v (B, Tk, E) separate contiguous tensors """
    B, Tq, E = q.data.shape
    Tk = k.data.shape[1]
    dh = E // heads
    o = K.mha(q.data, k.data, v.data, heads, scale, key_len=key_len)
    out = Var(o, q.requires_grad or k.requires_grad or v.requires_grad)
    if not out.requires_grad:
        return out

    def backward():
        if out.grad is None:
            return
        dO = out.grad
        dev = dO.device
        if _FUSED_ATTN_BWD and dh == 64 and dO.is_contiguous():
            # one fused backward (csrc/attn_bwd.hip): scores and probabilities stay on the chip
            stats = torch.empty((B, heads, Tq, 2), device=dev, dtype=torch.float32)
            dQ, dK, dV = torch.empty_like(q.data), torch.empty_like(k.data), torch.empty_like(v.data)
            K._check(_L().tocvp_attn_bwd_f32(_p(q.data), _p(k.data), _p(v.data), _p(out.data), _p(dO), _p(dQ),
                                             _p(dK), _p(dV), _p(stats), _p(key_len), B, heads, Tq, Tk, E,
                                             float(scale), _s()), "tocvp_attn_bwd_f32")
            accumulate(q, dQ)
            accumulate(k, dK)
            accumulate(v, dV)
            return
        hb = (B, heads)
        sQ, sK = (Tq * E, dh), (Tk * E, dh)
        sS = (heads * Tq * Tk, Tq * Tk)
        S = torch.empty((B, heads, Tq, Tk), device=dev, dtype=torch.float32)
        bmm(q.data, k.data, S, Tq, Tk, dh, E, E, Tk, transB=True, batch=hb, sA=sQ, sB=sK, sC=sS)
        K._check(_L().tocvp_softmax_rows_f32(_p(S), _p(S), B * heads * Tq, Tk, float(scale), _p(key_len),
                                             heads * Tq, _s()), "tocvp_softmax_rows_f32")
        if v.requires_grad:                               # dV = P^T dO
            dV = torch.empty_like(v.data)
            bmm(S, dO, dV, Tk, dh, Tq, Tk, E, E, transA=True, batch=hb, sA=sS, sB=sQ, sC=sK)
            accumulate(v, dV)
        if q.requires_grad or k.requires_grad:
            dP = torch.empty_like(S)                      # dP = dO V^T, then dS in place
            bmm(dO, v.data, dP, Tq, Tk, dh, E, E, Tk, transB=True, batch=hb, sA=sQ, sB=sK, sC=sS)
            K._check(_L().tocvp_softmax_bwd_f32(_p(S), _p(dP), _p(dP), B * heads * Tq, Tk, float(scale), _s()),
                     "tocvp_softmax_bwd_f32")
            if q.requires_grad:                           # dQ = dS K
                dQ = torch.empty_like(q.data)
                bmm(dP, k.data, dQ, Tq, dh, Tk, Tk, E, E, batch=hb, sA=sS, sB=sK, sC=sQ)
                accumulate(q, dQ)
            if k.requires_grad:                           # dK = dS^T Q
                dK = torch.empty_like(k.data)
                bmm(dP, q.data, dK, Tk, dh, Tq, Tk, E, E, transA=True, batch=hb, sA=sS, sB=sQ, sC=sK)
                accumulate(k, dK)
    tape.record(backward)
    return out


def _dropout_raw(x, r, p):
    y = torch.empty_like(x)
    K._check(_L().tocvp_dropout_f32(_p(x), _p(r), _p(y), x.numel(), float(p), _s()), "tocvp_dropout_f32")
    return y


def dropout(tape, x, p, generator=None, sample=None):
    """ nn.Dropout(p) in training mode; ``sample`` (uniform [0,1), same shape) overrides the generator """
    if p <= 0.0:
        return x
    r = sample if sample is not None else torch.rand(x.data.shape, device=x.data.device, generator=generator)
    out = Var(_dropout_raw(x.data, r, p), x.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is not None:
                accumulate(x, _dropout_raw(out.grad, r, p))
        tape.record(backward)
    return out


def attention_unfused(tape, q, k, v, heads, scale, key_len=None, p_drop=0.0, generator=None, sample=None):
    """
    Attention with dropout on the probabilities (nn.MultiheadAttention in training mode): the scores are
    materialised (B, H, Tq, Tk), so this is for short sequences (the caption encoder).
    """
    B, Tq, E = q.data.shape
    Tk = k.data.shape[1]
    dh = E // heads
    dev = q.data.device
    hb = (B, heads)
    sQ, sK = (Tq * E, dh), (Tk * E, dh)
    sS = (heads * Tq * Tk, Tq * Tk)
    P = torch.empty((B, heads, Tq, Tk), device=dev, dtype=torch.float32)
    bmm(q.data, k.data, P, Tq, Tk, dh, E, E, Tk, transB=True, batch=hb, sA=sQ, sB=sK, sC=sS)
    K._check(_L().tocvp_softmax_rows_f32(_p(P), _p(P), B * heads * Tq, Tk, float(scale), _p(key_len),
                                         heads * Tq, _s()), "tocvp_softmax_rows_f32")
    r = None
    Pd = P
    if p_drop > 0.0:
        r = sample if sample is not None else torch.rand(P.shape, device=dev, generator=generator)
        Pd = _dropout_raw(P, r, p_drop)
    o = torch.empty((B, Tq, E), device=dev, dtype=torch.float32)
    bmm(Pd, v.data, o, Tq, dh, Tk, Tk, E, E, batch=hb, sA=sS, sB=sK, sC=sQ)
    out = Var(o, q.requires_grad or k.requires_grad or v.requires_grad)
    if not out.requires_grad:
        return out

    def backward():
        if out.grad is None:
            return
        dO = out.grad
        if v.requires_grad:
            dV = torch.empty_like(v.data)
            bmm(Pd, dO, dV, Tk, dh, Tq, Tk, E, E, transA=True, batch=hb, sA=sS, sB=sQ, sC=sK)
            accumulate(v, dV)
        if q.requires_grad or k.requires_grad:
            dP = torch.empty_like(P)
            bmm(dO, v.data, dP, Tq, Tk, dh, E, E, Tk, transB=True, batch=hb, sA=sQ, sB=sK, sC=sS)
            if r is not None:
                dP = _dropout_raw(dP, r, p_drop)
            K._check(_L().tocvp_softmax_bwd_f32(_p(P), _p(dP), _p(dP), B * heads * Tq, Tk, float(scale), _s()),
                     "tocvp_softmax_bwd_f32")
            if q.requires_grad:
                dQ = torch.empty_like(q.data)
                bmm(dP, k.data, dQ, Tq, dh, Tk, Tk, E, E, batch=hb, sA=sS, sB=sK, sC=sQ)
                accumulate(q, dQ)
            if k.requires_grad:
                dK = torch.empty_like(k.data)
                bmm(dP, q.data, dK, Tk, dh, Tq, Tk, E, E, transA=True, batch=hb, sA=sS, sB=sQ, sC=sK)
                accumulate(k, dK)
    tape.record(backward)
    return out


def add(tape, a, b):
    y = a.data.clone()
    axpby(b.data, y, 1.0, 1.0)
    out = Var(y, a.requires_grad or b.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            if a.requires_grad and b.requires_grad:
                accumulate(a, out.grad.clone())
                accumulate(b, out.grad)
            else:
                accumulate(a if a.requires_grad else b, out.grad)
        tape.record(backward)
    return out


def add_position_rows(tape, x, table, index):
    """ x (B, w, ...rest) + table[index[pos]] broadcast over B and the inner axes; table (P, E), E = last axis """
    B, w = x.data.shape[:2]
    E = x.data.shape[-1]
    key = (tuple(int(i) for i in index), str(x.data.device))
    idx = _INDEX_CACHE.get(key)            # host -> device copies are not allowed while a graph is captured
    if idx is None:
        idx = _INDEX_CACHE[key] = torch.as_tensor(list(key[0]), device=x.data.device, dtype=torch.int64)
    rows = table.data.index_select(0, idx)                             # (w, E) gather (data movement)
    shape = [1, w] + [1] * (x.data.dim() - 3) + [E]
    addend = rows.reshape(shape).expand_as(x.data).contiguous()
    y = x.data.clone()
    axpby(addend, y, 1.0, 1.0)
    out = Var(y, x.requires_grad or table.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            if table.requires_grad:
                if table.grad is None:
                    table.grad = torch.zeros_like(table.data)
                g = out.grad.reshape(B, w, -1, E).permute(1, 0, 2, 3).contiguous()   # (w, B, inner, E)
                for pos in range(w):
                    colsum(g[pos].reshape(-1, E), out=table.grad[int(index[pos])], acc=True)
            if x.requires_grad:
                accumulate(x, out.grad)
        tape.record(backward)
    return out


def stack_frames(tape, frames):
    """ list of Vars (B, K, D) -> Var (B, w, K, D) """
    y = torch.stack([f.data for f in frames], dim=1).contiguous()
    out = Var(y, any(f.requires_grad for f in frames))
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            for i, f in enumerate(frames):
                if f.requires_grad:
                    accumulate(f, out.grad[:, i].contiguous())
        tape.record(backward)
    return out


def take_frame(tape, x, i):
    """ x (B, w, K, E) -> frame i (B, K, E) """
    y = x.data[:, i].contiguous()
    out = Var(y, x.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            g = torch.zeros_like(x.data)
            g[:, i] = out.grad
            accumulate(x, g)
        tape.record(backward)
    return out


def take_last_tokens(tape, x, n):
    """ x (B, T, E) -> its last n tokens (B, n, E) (contiguous copy); the gradient of the other tokens is 0 """
    B, T, E = x.data.shape
    out = Var(x.data[:, T - n:].contiguous(), x.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            g = torch.zeros((B, T, E), device=out.grad.device, dtype=torch.float32)
            g[:, T - n:] = out.grad
            accumulate(x, g)
        tape.record(backward)
    return out


def transpose12(tape, x):
    """ x (B, a, b, E) -> (B, b, a, E) contiguous copy (data movement); the gradient takes the same way back """
    out = Var(x.data.transpose(1, 2).contiguous(), x.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is not None:
                accumulate(x, out.grad.transpose(1, 2).contiguous())
        tape.record(backward)
    return out


def reshape(tape, x, shape):
    """ view of a contiguous Var under another shape; the gradient is the same storage reshaped back """
    out = Var(x.data.reshape(shape), x.requires_grad)
    if out.requires_grad:
        tape.record(lambda: accumulate(x, out.grad.reshape(x.data.shape)) if out.grad is not None else None)
    return out


def embedding(tape, ids, table):
    y = K.embedding(ids, table.data)
    out = Var(y, table.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            if table.grad is None:
                table.grad = torch.zeros_like(table.data)
            D = table.data.shape[1]
            K._check(_L().tocvp_embedding_bwd_f32(_p(ids), _p(out.grad), _p(table.grad), ids.numel(), D, _s()),
                     "tocvp_embedding_bwd_f32")
        tape.record(backward)
    return out


def mask_rows(tape, x, keep):
    """ zero the rows of x (..., E) where ``keep`` (..., bool) is False (caption padding) """
    y = x.data.clone()
    y.masked_fill_(~keep.unsqueeze(-1), 0.0)
    out = Var(y, x.requires_grad)
    if out.requires_grad:
        def backward():
            if out.grad is None:
                return
            g = out.grad.clone()
            g.masked_fill_(~keep.unsqueeze(-1), 0.0)
            accumulate(x, g)
        tape.record(backward)
    return out


def mse(tape, pred, target, weight=1.0):
    """ weight * mean((pred - target)^2); returns the python float and seeds pred.grad """
    n = pred.data.numel()
    nblocks = min(1024, (n + 255) // 256)
    part = torch.empty(nblocks, device=pred.data.device, dtype=torch.float32)
    dp = torch.empty_like(pred.data) if pred.requires_grad else None
    tgt = target if target.is_contiguous() else target.contiguous()
    K._check(_L().tocvp_mse_f32(_p(pred.data), _p(tgt), _p(part), nblocks, _p(dp), n, 2.0 * weight / n, _s()),
             "tocvp_mse_f32")
    total = colsum(part.reshape(nblocks, 1))
    if dp is not None:
        tape.record(lambda: accumulate(pred, dp))
    return total, float(weight) / n                   # (1,) device tensor (no host sync here) and its scale
