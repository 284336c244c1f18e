#!/usr/bin/env python
"""
Numerics probe (CPU, torch): would the SAVi decoder's three 64 -> 64 5x5 convolutions hold the parity bar
if they were evaluated as Winograd F(2x2, 5x5) products (36 position products per 2 x 2 outputs instead of
100 tap products: 2.78 x fewer matrix products) in the split-fp16 arithmetic of the present kernels?

Compares, on slots of a committed golden fixture and the synthetic weight families:
    fp64 direct (truth) | fp32 direct | split-fp16 direct (simulated) | split-fp16 Winograd (simulated, several
    point sets) -- rendered frames, slot masks and the per-pixel winning slot.

    python scripts/probes/winograd_numerics.py [family] [frames] [only1d]

(third argument: only the 1-D forms -- F(2, 5) / F(4, 5) along y nested with the five direct taps along x, the second of which
csrc/conv_wino.hip implements -- with scales picked from the data and with the kernel's fixed ones; the matrix core's flush of
fp16 subnormals is modelled, FLUSH below.)
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from textocvp_amd import synth                                   # noqa: E402


# ---- Cook-Toom matrices for F(m, r) from a list of alpha - 1 finite points (+ infinity) -------------------
def poly_mul(a, b):
    out = [Fraction(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def cook_toom(m, r, pts):
    al = m + r - 1
    pts = [Fraction(p) for p in pts]
    assert len(pts) == al - 1
    AT = [[Fraction(0)] * al for _ in range(m)]
    G = [[Fraction(0)] * r for _ in range(al)]
    BT = [[Fraction(0)] * al for _ in range(al)]
    for j, p in enumerate(pts):
        N = Fraction(1)
        for l, q in enumerate(pts):
            if l != j:
                N *= (p - q)
        for i in range(m):
            AT[i][j] = p ** i
        for k in range(r):
            G[j][k] = p ** k / N
        poly = [Fraction(1)]
        for l, q in enumerate(pts):
            if l != j:
                poly = poly_mul(poly, [-q, Fraction(1)])
        for n, c in enumerate(poly):
            BT[j][n] = c
    AT[m - 1][al - 1] = Fraction(1)
    G[al - 1][r - 1] = Fraction(1)
    poly = [Fraction(1)]
    for q in pts:
        poly = poly_mul(poly, [-q, Fraction(1)])
    for n, c in enumerate(poly):
        BT[al - 1][n] = c
    # the finite rows need the sign / scale that makes the identity exact: solve for a per-row factor numerically
    A_ = np.array(AT, dtype=np.float64)
    G_ = np.array(G, dtype=np.float64)
    B_ = np.array(BT, dtype=np.float64)
    # identity: sum_j AT[i, j] G[j, k] BT[j, n] s_j = delta(n == i + k); solve s by least squares
    rows, rhs = [], []
    for i in range(m):
        for k in range(r):
            for n in range(al):
                rows.append(A_[i, :] * G_[:, k] * B_[:, n])
                rhs.append(1.0 if n == i + k else 0.0)
    s, res, *_ = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)
    assert np.allclose(np.array(rows) @ s, rhs, atol=1e-9), "Cook-Toom identity does not hold"
    B_ = B_ * s[:, None]
    return A_, G_, B_


FLUSH = True


# ---- split-fp16 arithmetic, simulated ----------------------------------------------------------------------
def split16(x, scale):
    xs = (x.double() * scale).float()
    hi = xs.half()
    lo = (xs - hi.float()).half()
    hi, lo = hi.float(), lo.float()
    if FLUSH:                                   # the matrix core flushes fp16 subnormals
        hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi, lo


def mm3(a, b, sa, sb):
    """ fp32-accumulated a @ b with both operands as two fp16 planes, three of the four plane products """
    ah, al_ = split16(a, sa)
    bh, bl = split16(b, sb)
    return ((ah @ bh) + ((ah @ bl) + (al_ @ bh))) / (sa * sb)


def pow2_scale(x, limit=30000.0):
    m = float(x.abs().max())
    return 2.0 ** np.floor(np.log2(limit / max(m, 1e-30)))


def conv_direct_split(x, w, b):
    """ 5x5 same conv as 25 tap GEMMs in split-fp16 (x (n, C, H, W) fp32) """
    n, C, H, W = x.shape
    xp = F.pad(x, (2, 2, 2, 2))
    sa, sb = pow2_scale(x), pow2_scale(w)
    acc = torch.zeros((n * H * W, w.shape[0]), dtype=torch.float32)
    for dy in range(5):
        for dx in range(5):
            a = xp[:, :, dy:dy + H, dx:dx + W].permute(0, 2, 3, 1).reshape(-1, C)
            acc += mm3(a, w[:, :, dy, dx].t().contiguous(), sa, sb)
    y = acc.reshape(n, H, W, -1).permute(0, 3, 1, 2) + b[None, :, None, None]
    return y, sa


def conv_winograd_split(x, w, b, mats, split=True):
    """ F(2x2, 5x5) with fp32 transforms and split-fp16 position products """
    AT, G, BT = mats
    n, C, H, W = x.shape
    al = BT.shape[0]
    U = torch.einsum("ak,ockl,bl->aboc", torch.from_numpy(G), w.double(), torch.from_numpy(G))       # (6,6,O,C) fp64
    xp = F.pad(x, (2, 2, 2, 2))
    # tiles: (n, C, H/2, W/2, 6, 6)
    t = xp.unfold(2, al, 2).unfold(3, al, 2)
    BTf = torch.from_numpy(BT).float()
    V = torch.einsum("ai,nchwij,bj->abnhwc", BTf, t, BTf)                                            # fp32 transform
    th, tw = V.shape[3], V.shape[4]
    sa = pow2_scale(V)
    sb = pow2_scale(U)
    M = torch.empty((al, al, n * th * tw, w.shape[0]), dtype=torch.float32)
    for a in range(al):
        for bb in range(al):
            v = V[a, bb].reshape(-1, C)
            u = U[a, bb].t().contiguous().float() if not split else U[a, bb].t().contiguous()
            M[a, bb] = mm3(v, u, sa, sb) if split else v @ u.float()
    ATf = torch.from_numpy(AT).float()
    Y = torch.einsum("ia,abmo,jb->mijo", ATf, M, ATf)                                                # (tiles, 2, 2, O)
    Y = Y.reshape(n, th, tw, 2, 2, -1).permute(0, 5, 1, 3, 2, 4).reshape(n, -1, H, W)
    return Y + b[None, :, None, None], sa, float(V.abs().max()) / max(float(x.abs().max()), 1e-30)


def conv_winograd_1d_split(x, w, b, mats, m, sa=None, sb=None):
    """ F(m, 5) along y nested with the 5 direct taps along x: (m + 4) * 5 products per m outputs; fp32 transforms,
    split-fp16 products accumulated in fp32 over (dx, c) per transform row, fp32 output transform """
    AT, G, BT = mats
    n, C, H, W = x.shape
    al = BT.shape[0]
    U = torch.einsum("ak,ockl->aloc", torch.from_numpy(G), w.double())                               # (al, 5, O, C) fp64
    xp = F.pad(x, (2, 2, 2, 2))
    t = xp.unfold(2, al, m)                                                                          # (n, C, H/m, W+4, al)
    BTf = torch.from_numpy(BT).float()
    V = torch.einsum("ai,nchxi->anhxc", BTf, t)                                                      # (al, n, H/m, W+4, C)
    growth = float(V.abs().max()) / max(float(x.abs().max()), 1e-30)
    sa = pow2_scale(V) if sa is None else sa
    sb = pow2_scale(U) if sb is None else sb
    th = V.shape[2]
    M = torch.zeros((al, n * th * W, w.shape[0]), dtype=torch.float32)
    for a in range(al):
        sb_a = pow2_scale(U[a], 16384.0) if sb == "per-row" else sb
        for dx in range(5):
            v = V[a, :, :, dx:dx + W].reshape(-1, C)
            M[a] += mm3(v, U[a, dx].t().contiguous(), sa, sb_a)
    ATf = torch.from_numpy(AT).float()
    Y = torch.einsum("ia,amo->mio", ATf, M)                                                          # (n*th*W, m, O)
    Y = Y.reshape(n, th, W, m, -1).permute(0, 4, 1, 3, 2).reshape(n, -1, H, W)
    return Y + b[None, :, None, None], sa, growth


def decode(sd, slots, conv, dtype=torch.float32, log=None):
    import oracle.slot_rollout_oracle as O
    Bp, K, D = slots.shape
    pos = O.soft_pos_embed(sd["decoder_pos_embedding.projection.weight"].float(),
                           sd["decoder_pos_embedding.projection.bias"].float(), (64, 64)).to(dtype)
    x = (slots.to(dtype).reshape(Bp * K, 1, 1, D) + pos[None]).permute(0, 3, 1, 2)
    i = 0
    while f"decoder.decoder.{i}.block.0.weight" in sd:
        w, b = sd[f"decoder.decoder.{i}.block.0.weight"].to(dtype), sd[f"decoder.decoder.{i}.block.0.bias"].to(dtype)
        if i == 0 and conv is not None:                   # only the three 64 -> 64 layers are under test
            x = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2)).float()
        elif conv is None:
            x = torch.relu(F.conv2d(x, w, b, padding=2))
        else:
            y = conv(x, w, b)
            if log is not None:
                log.append(y[1:])
            x = torch.relu(y[0])
        i += 1
    w, b = sd[f"decoder.decoder.{i}.weight"].to(dtype), sd[f"decoder.decoder.{i}.bias"].to(dtype)
    decode.last_hidden = x
    if conv is not None:
        x, w, b = x.double(), w.double(), b.double()
    y = F.conv2d(x, w, b, padding=1).reshape(Bp, K, 4, 64, 64)
    recons, alpha = y[:, :, :3], y[:, :, 3:]
    masks = torch.softmax(alpha, dim=1)
    return (recons * masks).sum(dim=1), masks


def main():
    family = sys.argv[1] if len(sys.argv) > 1 else "damped"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    torch.manual_seed(0)
    torch.set_num_threads(8)
    man = json.load(open(os.path.join(ROOT, "tests/golden/state_dict_manifest.json")))["SAVi"]
    sd = synth.synth_state_dict(man, prefix="savi.", family=family)
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    g = np.load(os.path.join(ROOT, "tests/golden/e2e_c2.npz"))
    slots = torch.from_numpy(g["pred_slots"][0, ::max(1, 19 // frames)][:frames]).float()           # (frames, 30, 128)

    truth_img, truth_m = decode(sd, slots, None, torch.float64)
    truth_h = decode.last_hidden
    win = truth_m.argmax(dim=1)
    top2 = truth_m.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).flatten()
    print(f"family {family}, {frames} frames x 30 slots; smallest winning margins of the fp64 masks: "
          f"{np.sort(margin.numpy())[:5]}")

    def report(name, img, m, extra=""):
        e_img = float((img.double() - truth_img).abs().max())
        e_m = float((m.double() - truth_m).abs().max())
        flips = int((m.argmax(dim=1) != win).sum())
        e_h = float((decode.last_hidden.double() - truth_h).abs().max() / truth_h.abs().max())
        print(f"{name:72s} frames {e_img:9.2e}  masks {e_m:9.2e}  last hidden (rel. to max) {e_h:9.2e}  "
              f"winning-slot flips {flips:5d} / {win.numel()} {extra}")

    report("fp32 direct (torch)", *decode(sd, slots, None, torch.float32))
    report("split-fp16 direct (3 plane products)", *decode(sd, slots, conv_direct_split))

    point_sets = {
        "0 +-1 +-2 inf": [0, 1, -1, 2, -2],
        "0 +-1 +-1/2 inf": [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)],
        "0 +-1/2 +-2 inf": [0, Fraction(1, 2), Fraction(-1, 2), 2, -2],
        "0 +-1 +2 -1/2 inf": [0, 1, -1, 2, Fraction(-1, 2)],
        "0 +-1/2 +-3/2 inf": [0, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 2), Fraction(-3, 2)],
    }
    H_ = Fraction(1, 2)
    sets_1d = {
        (2, "0 +-1 +-2 inf"): [0, 1, -1, 2, -2],
        (2, "0 +-1 +-1/2 inf"): [0, 1, -1, H_, -H_],
        (4, "0 +-1 +-2 +-1/2 inf"): [0, 1, -1, 2, -2, H_, -H_],
        (4, "0 +-1 +-1/2 +-3/2 inf"): [0, 1, -1, H_, -H_, 3 * H_, -3 * H_],
        (4, "0 +-1/2 +-1 +-1/4... (0, +-1, +-1/2, +-1/4)"): [0, 1, -1, H_, -H_, H_ / 2, -H_ / 2],
    }
    for (m, name), pts in sets_1d.items():
        mats = cook_toom(m, 5, pts)
        rowsum = np.abs(mats[2]).sum(axis=1).max()
        gsum = np.abs(mats[1]).sum(axis=1).max()
        for fixed in (False, 1.0, 4.0, 16.0):
            log = []
            kw = dict(sa=fixed * 2.0 ** 8 / 2.0 ** np.ceil(np.log2(rowsum)), sb='per-row') if fixed else {}
            img, mk = decode(sd, slots, lambda x, w, b: conv_winograd_1d_split(x, w, b, mats, m, **kw), log=log)
            growth = " ".join(f"{l[1]:.1f}x" for l in log)
            report(f"split-fp16 1-D F({m},5) [{name}]{' fixed scales' if fixed else ''}"[:70], img, mk,
                   f" growth {growth}; |B^T| row sum {rowsum:.2f}, |G| row sum {gsum:.2f}" + (f" sa {kw['sa']}" if fixed else ""))
    if len(sys.argv) > 3:
        return
    for name, pts in point_sets.items():
        mats = cook_toom(2, 5, pts)
        log = []
        img, m = decode(sd, slots, lambda x, w, b: conv_winograd_split(x, w, b, mats), log=log)
        growth = " ".join(f"{l[1]:.0f}x" for l in log)
        report(f"split-fp16 Winograd F(2,5) [{name}]", img, m, f" input growth per layer {growth}")
        img, m = decode(sd, slots, lambda x, w, b: conv_winograd_split(x, w, b, mats, split=False))
        report(f"      fp32  Winograd F(2,5) [{name}]", img, m)


if __name__ == "__main__":
    main()
