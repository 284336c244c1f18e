"""
Rollout evaluation harness: this repo's counterpart of the reference's ``Evaluator.forward_eval``
(05_evaluate_predictor.py:53-104) and of the nn.DataParallel wrapping in base/baseEvaluator.py
(:142-145, :168-171), re-designed for one process per GPU.

  * ``forward_eval``  -- the three module calls + reshape + clamp, identical glue to the reference.
  * ``forward_eval_decomp`` -- the decomposition-only evaluation (03_evaluate_decomp_model.py:22-46):
    ``model(x, num_imgs)`` with the default ``mode`` / ``decode=True``, clamp, metric tracker.
  * ``shard_batches`` -- whole reference batches are dealt round-robin to ranks (batch j -> rank
    j mod W): caption padding is per batch and leaks into predictions, so only
    batch-preserving sharding reproduces single-process results.
  * ``gather_metrics`` -- ONE all-gather of the per-sequence (N_r, P) metric tensors at the end
    (RCCL over xGMI on GPUs, gloo in the CPU tests); no communication inside the rollout.
"""

import os

import torch
import torch.distributed as dist

__all__ = ["forward_eval", "forward_eval_decomp", "shard_batches", "gather_metrics"]


_SIDE_STREAMS = {}
_ENCODE_STREAMS = {}
_OVERLAP_ENCODE = os.environ.get("TOCVP_OVERLAP_ENCODE", "1") != "0"
_OVERLAP_ENCODE_MIN_BATCH = int(os.environ.get("TOCVP_OVERLAP_ENCODE_MIN_BATCH", "1"))


def _encode_stream(device):
    key = (device.type, device.index)
    if key not in _ENCODE_STREAMS:
        _ENCODE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _ENCODE_STREAMS[key]



def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


# The rollout's chain of short kernels runs on a HIGH-priority stream (TOCVP_ROLLOUT_PRIORITY=0: the caller's stream), so that its
# workgroups are dispatched ahead of the decoder's 16320-workgroup grids whenever a CU frees up.  Round 4, two alternations on
# one box: B = 32 3449 / 3494 -> 3541 / 3556 frames/s (+2 %), B = 128 4008 / 4014 -> 4034 / 4029 (+0.5 %), B = 8 neutral
# (host-bound).  (Round 3 measured the same idea neutral at B = 128 with the two-GEMM MLPs.)  Same kernels, same results.
_ROLLOUT_PRIORITY = os.environ.get("TOCVP_ROLLOUT_PRIORITY", "1") != "0"
_ROLLOUT_STREAMS = {}


def _rollout_stream(device):
    key = (device.type, device.index)
    if key not in _ROLLOUT_STREAMS:
        _ROLLOUT_STREAMS[key] = torch.cuda.Stream(device=device, priority=-1)
    return _ROLLOUT_STREAMS[key]


def _reads_context_only(predictor, num_context):
    """
    True when the rollout provably reads ``slot_history[:, :num_context]`` and nothing behind it, so the frames behind
    the context may be decomposed later.  A wrapper with ``teacher_force`` on reads ``slot_history[:, num_context + t]``
    (predictor_wrapper.py:74-82 -- the reference applies the config value in eval mode too), and a wrapper whose own
    ``num_context`` exceeds the evaluator's reads more context frames than the cut would hand it.  Anything that is not a
    recognisable PredictorWrapper gets the full history.
    """
    wrapper = getattr(predictor, "module", predictor)
    params = getattr(wrapper, "exp_params", None)
    if not isinstance(params, dict) or "prediction_params" not in params:
        return False
    if params["prediction_params"].get("teacher_force", False):      # what _rollout re-reads on every call
        return False
    return int(getattr(wrapper, "num_context", num_context + 1)) <= num_context


@torch.no_grad()
def forward_eval(decomp_model, predictor, videos, num_context, num_preds, overlap_decode=None,
                 _calibrating=False, **others):
    """
    videos (B, L, C, H, W) in [0,1]; ``others`` carries caption_tokens / caption_lengths
    (and optionally init_noise).  Returns dict(slot_history, pred_slots, pred_imgs, targets, masks,
    recons, recons_imgs).

    overlap_decode (default on; env TOCVP_OVERLAP_DECODE=0 turns it off): frame t is decoded on a SECOND HIP stream
    as soon as rollout step t is enqueued.  The rollout is a chain of short dependent kernels that
    cannot fill 256 CUs (especially while the window is short); the MFMA-bound decoder convolutions
    of already-predicted frames run in the gaps.  Same kernels, same arithmetic, same results.
    """
    B, L, C, H, W = videos.shape
    if L < num_context + num_preds:
        raise ValueError(f"Seq. length {L} smaller that {num_context = } + {num_preds = }")
    if not _calibrating and (getattr(decomp_model, "_range_unchecked", False)
                             or getattr(predictor, "_range_unchecked", False)):
        # freshly loaded weights: one checked pass on this batch; a module whose operands leave the
        # fp16-plane range is moved to range-free arithmetic (never silent saturation)
        import warnings
        from .setup_model import calibrate_precision
        changed = calibrate_precision(decomp_model, predictor, videos, num_context, num_preds, **others)
        if changed:
            warnings.warn(f"textocvp_amd: operands outside the fp16-plane range, arithmetic changed: {changed}")
    num_slots, slot_dim = decomp_model.num_slots, decomp_model.slot_dim
    if overlap_decode is None:
        # on by default at every batch size: same kernels, same arithmetic, bit-identical results
        # (tests: test_decode_overlap_is_bit_identical, test_bench_shape_b128_*).  Below ~96 sequences the rollout
        # cannot fill the chip and the gain is large; at B=128 it is +2-3 %.
        # TOCVP_OVERLAP_DECODE=0 runs the serial order (what bench.py's per-kernel attribution pass uses).
        overlap_decode = os.environ.get("TOCVP_OVERLAP_DECODE", "1") != "0"
    # The rollout needs the slots of the context frames only; the frames behind them are decomposed for the returned
    # slot_history alone.  With the overlap on (and a model that can cut its decomposition: SAVi.decomp_frames) they
    # are decomposed on a third stream while the rollout and the decoder already run -- the same kernels on the same
    # images, bit-identical (TOCVP_OVERLAP_ENCODE=0: the whole decomposition first, as the reference orders it).
    core = getattr(decomp_model, "module", decomp_model)
    n_all = num_context + num_preds
    rest = None
    if (overlap_decode and videos.is_cuda and _OVERLAP_ENCODE and hasattr(core, "decomp_frames") and num_preds > 0
            and not getattr(core, "_range_unchecked", False) and B >= _OVERLAP_ENCODE_MIN_BATCH
            and _reads_context_only(predictor, num_context)):
        main = torch.cuda.current_stream()
        enc = _encode_stream(videos.device)
        predicted = core.initializer(batch_size=B, **others)
        first, state = core.decomp_frames(videos, 0, num_context, predicted)
        from . import kernels as K
        # (B, num_context + num_preds, K, D): the context frames now, the later frames from the third stream (no cat)
        hist_full = torch.empty((B, n_all) + tuple(first[0].shape[1:]), device=first[0].device, dtype=torch.float32)
        for i_, s_ in enumerate(first):
            K.copy_strided(s_, hist_full[:, i_])
        slot_ctx = K.contiguous(hist_full[:, :num_context])           # (B, num_context, K, D): what the rollout reads
        enc.wait_stream(main)
        videos.record_stream(enc)
        state.record_stream(enc)
        hist_full.record_stream(enc)
        with torch.cuda.stream(enc):
            later, _ = core.decomp_frames(videos, num_context, n_all, state)
            for i_, s_ in enumerate(later):
                K.copy_strided(s_, hist_full[:, num_context + i_])
            rest = hist_full
        slot_history = slot_ctx                                       # what the rollout reads
    else:
        out_model = decomp_model(mode="decomp", x=videos, num_imgs=n_all, decode=False, **others)
        slot_history = out_model["slot_history"]
    # The decoder's tail kernel writes straight into the results the reference reshapes out of ONE decode call
    # (05_evaluate_predictor.py:88-96): recons_imgs / recons / masks as (B * P, ...) with row b * P + t, and the clamped
    # frames pred_imgs (B, P, C, H, W) from the same kernel -- no stack, no copy, no elementwise clamp.  Models whose
    # decode has no ``out`` placement (ExtendedDINOSAUR) keep the stack / clamp form.
    placed = slot_history.is_cuda and getattr(core, "decode_accepts_out", False)
    if placed:
        dev = slot_history.device
        F_ = B * num_preds
        recons_imgs = torch.empty((F_, C, H, W), device=dev, dtype=torch.float32)
        recons = torch.empty((F_, num_slots, C, H, W), device=dev, dtype=torch.float32)
        masks = torch.empty((F_, num_slots, 1, H, W), device=dev, dtype=torch.float32)
        pred_imgs = torch.empty((B, num_preds, C, H, W), device=dev, dtype=torch.float32)
        full = (recons_imgs, recons, masks, pred_imgs.view(F_, C, H, W))
    if not (overlap_decode and slot_history.is_cuda):
        pred_slots = predictor(slot_history, **others)
        flat = pred_slots.reshape(B * num_preds, num_slots, slot_dim)
        if placed:
            decomp_model(mode="decode", slots=flat, out=full)
        else:
            out_dec = decomp_model(mode="decode", slots=flat)
            pred_imgs = out_dec["recons_imgs"].view(B, num_preds, C, H, W).clamp(0, 1)
            masks, recons, recons_imgs = out_dec["masks"], out_dec.get("recons"), out_dec["recons_imgs"]
    else:
        main = torch.cuda.current_stream()
        side = _side_stream(slot_history.device)
        per_step = [None] * num_preds
        if placed:
            for t_ in full:
                t_.record_stream(side)
            step_views = [t_.view(B, num_preds, *t_.shape[1:]) for t_ in full]

        def decode_step(t, pred_t):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            pred_t.record_stream(side)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                if placed:
                    decomp_model(mode="decode", slots=pred_t, out=tuple(v[:, t] for v in step_views))
                else:
                    per_step[t] = decomp_model(mode="decode", slots=pred_t)
        side.wait_stream(main)                                    # decoder weights / caches / outputs are ready
        if _ROLLOUT_PRIORITY:
            hp = _rollout_stream(slot_history.device)
            hp.wait_stream(main)
            slot_history.record_stream(hp)
            with torch.cuda.stream(hp):
                pred_slots = predictor(slot_history, step_callback=decode_step, **others)
            main.wait_stream(hp)
            pred_slots.record_stream(main)
        else:
            pred_slots = predictor(slot_history, step_callback=decode_step, **others)
        main.wait_stream(side)
        if not placed:
            imgs = torch.stack([d["recons_imgs"] for d in per_step], dim=1)          # (B, P, C, H, W)
            masks = torch.stack([d["masks"] for d in per_step], dim=1)
            masks = masks.reshape(B * num_preds, *masks.shape[2:])
            recons = None
            if "recons" in per_step[0]:              # SAVi; ExtendedDINOSAUR returns recons_feats instead
                recons = torch.stack([d["recons"] for d in per_step], dim=1)
                recons = recons.reshape(B * num_preds, *recons.shape[2:])
            recons_imgs = imgs.reshape(B * num_preds, C, H, W)
            pred_imgs = imgs.clamp(0, 1)
    if rest is not None:                                              # join the decomposition of the later frames
        cur = torch.cuda.current_stream()
        cur.wait_stream(_encode_stream(videos.device))
        rest.record_stream(cur)
        slot_history = rest
    targets = videos[:, num_context:num_context + num_preds].to(pred_imgs.device)
    if targets.is_cuda and targets.dtype == torch.float32 and (C * H * W) % 4 == 0 and targets.data_ptr() % 16 == 0:
        from . import kernels as K
        targets = K.clamp01_rows(targets)                 # one pass over the row-strided slice, torch.clamp semantics
    else:
        targets = targets.clamp(0, 1)
    # recons / recons_imgs: SAVi.decode's per-slot and composited frames, (B*P, ...) and UNclamped
    return {"slot_history": slot_history, "pred_slots": pred_slots, "pred_imgs": pred_imgs,
            "targets": targets, "masks": masks, "recons": recons, "recons_imgs": recons_imgs}


class GraphedEval:
    """
    ``forward_eval`` replayed from a captured HIP graph: the small-batch form of the path.

    One sequence is ~2400 dependent kernel launches of a few microseconds each; enqueued from Python the host is the
    bottleneck (B = 1: 46 ms eager against 28 ms of device time).  The first call with a new input signature (shapes
    and dtypes of ``videos`` and of every tensor in ``others``) runs one eager pass on private copies of the inputs
    -- range calibration, weight splits, workspaces, the side stream -- then captures the same pass into a graph;
    every later call copies its inputs into those buffers and replays.  Same kernels in the same order on the same
    arithmetic: results are bit-identical to the eager call (tests/test_boundary_gpu.py).

    A graph holds raw pointers to the weights' derived forms (operand planes, fused tables): when a parameter or buffer
    of either model is replaced, moved or modified (``load_state_dict``, ``.to``, an optimiser step) every graph is
    dropped and the next call captures again.

    The returned tensors are the graph's own output buffers: they are overwritten by the next call with the same
    signature, so consume (or clone) them first.  ``epilogue(out) -> tensor or tuple`` (optional) runs inside the
    capture on the result dictionary, e.g. the fused PSNR / SSIM step; its value is returned under "epilogue".
    Non-tensor entries of ``others`` (the caption strings the reference also passes) only reach the capturing call.
    The decoder is not overlapped with the rollout by default here: under replay the serial order was the faster
    one at every batch measured (B = 1: 28.1 vs 32.7 ms, B = 8: 66.3 vs 67.2, B = 32: 185.0 vs 187.4).
    """

    def __init__(self, decomp_model, predictor, num_context, num_preds, overlap_decode=False, epilogue=None):
        self.decomp_model, self.predictor = decomp_model, predictor
        self.num_context, self.num_preds = num_context, num_preds
        self.overlap_decode, self.epilogue = overlap_decode, epilogue
        self._graphs, self._weights = {}, None

    def _weights_signature(self):
        sig = []
        for m in (self.decomp_model, self.predictor):
            sig += [(t.data_ptr(), t._version) for t in m.parameters()]
            sig += [(t.data_ptr(), t._version) for t in m.buffers()]
        return hash(tuple(sig))

    @staticmethod
    def _signature(videos, others):
        sig = [("videos", tuple(videos.shape), videos.dtype)]
        sig += [(k, tuple(v.shape), v.dtype) for k, v in sorted(others.items()) if torch.is_tensor(v)]
        return tuple(sig)

    def _run(self, videos, others):
        out = forward_eval(self.decomp_model, self.predictor, videos, self.num_context, self.num_preds,
                           overlap_decode=self.overlap_decode, **others)
        if self.epilogue is not None:
            out["epilogue"] = self.epilogue(out)
        return out

    def _with_init_noise(self, videos, others):
        """ The random slot initialiser draws one Gaussian per forward on the CPU generator (initializers.py:93 in the
        reference); a replay would freeze the capture's draw.  Draw it here, per call, exactly as the eager forward
        would, and hand it to the graph as an input. """
        if "init_noise" in others:
            return others
        model = getattr(self.decomp_model, "module", self.decomp_model)
        init = getattr(model, "initializer", None)
        if init is None or not hasattr(init, "slots_sigma"):
            return others
        noise = torch.randn((videos.shape[0], init.num_slots, init.slot_dim))
        return dict(others, init_noise=noise.to(videos.device))

    @torch.no_grad()
    def __call__(self, videos, **others):
        if not videos.is_cuda:
            raise ValueError("GraphedEval replays a HIP graph: the inputs must live on the GPU")
        others = self._with_init_noise(videos, others)
        weights = self._weights_signature()
        if weights != self._weights:
            self._graphs, self._weights = {}, weights
        key = self._signature(videos, others)
        entry = self._graphs.get(key)
        if entry is None:
            static_v = videos.clone()
            static_o = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in others.items()}
            self._run(static_v, static_o)                       # eager pass: calibration, caches, workspaces
            torch.cuda.synchronize(videos.device)
            static_v.copy_(videos)                              # new tensor versions: no cached caption projections
            for k, v in others.items():
                if torch.is_tensor(v):
                    static_o[k].copy_(v)
            graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: calls other threads make meanwhile (the RCCL watchdog polling its events
            # when a process group is alive) must not invalidate the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self._run(static_v, static_o)
            entry = (static_v, static_o, graph, out)
            self._graphs[key] = entry
        static_v, static_o, graph, out = entry
        static_v.copy_(videos)
        for k, v in others.items():
            if torch.is_tensor(v):
                static_o[k].copy_(v)
        graph.replay()
        return out


@torch.no_grad()
def forward_eval_decomp(decomp_model, videos, metric_tracker=None, **others):
    """
    Decomposition-only evaluation, the glue of the reference's ``Evaluator.forward_eval`` in
    03_evaluate_decomp_model.py:22-46: every frame of ``videos`` (B, L, C, H, W) is decomposed AND
    rendered (``mode`` and ``decode`` keep their defaults "decomp" / True, ``num_imgs = L``), the
    reconstruction is clamped to [0, 1] and, if a tracker is given, accumulated against the raw
    ``videos`` (:34-44).  ``others`` is what ``unwrap_batch_data`` yields (captions etc.: ignored by the
    decomposition model) plus, optionally, ``init_noise``.
    Returns the model's dictionary (recons_objs (B, L, K, C, H, W), masks (B, L, K, 1, H, W),
    slot_history (B, L, K, D), recons_imgs UNclamped) with the clamped frames under "recons_clamped".
    """
    if videos.dim() != 5:
        raise ValueError(f"videos with {videos.shape = }, but it must be (B, L, C, H, W)")
    out_model = decomp_model(x=videos, num_imgs=videos.shape[1], **others)
    recons_imgs = out_model.get("recons_imgs").clamp(0, 1)
    if metric_tracker is not None:
        metric_tracker.accumulate(preds=recons_imgs, targets=videos.to(recons_imgs.device))
    out = dict(out_model)
    out["recons_clamped"] = recons_imgs
    return out


def shard_batches(num_batches, rank, world_size):
    """ indices of the reference batches owned by ``rank`` (round-robin, batch-preserving) """
    return list(range(rank, num_batches, world_size))


def gather_metrics(local, group=None):
    """
    local: (N_r, P) float tensor of per-sequence metrics on this rank (N_r may differ per rank).
    Returns the (sum_r N_r, P) tensor in rank order on every rank, using a single padded
    all-gather (+ one tiny all-gather of the row counts).
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    home = local.device
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        local = local.cpu()            # single-GPU rehearsal: gloo moves host copies
    n_local = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    padded = torch.zeros((n_max,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    padded[:local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).to(home)
