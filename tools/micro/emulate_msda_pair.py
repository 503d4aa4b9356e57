"""Lane-level emulation (numpy, CPU) of `msda_bwd_pair_kernel` (vidar_amd/csrc/msda.hip): the same
workgroup / wave / half-wave index arithmetic, LDS slot layout, corner exchange and merge rule,
executed one workgroup at a time, checked against autograd of the gather formula.  Catches
transcription errors of the index math before the kernel sees a GPU; it says nothing about timing.
    python tools/micro/emulate_msda_pair.py"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import msda as M  # noqa: E402  (checker only)

K_CH, K_PHEADS = 32, 4


def corners(x, y, Hl, Wl, base, row_stride):
    h0, w0 = int(np.floor(y)), int(np.floor(x))
    h1, w1 = h0 + 1, w0 + 1
    lh, lw = np.float32(y - h0), np.float32(x - w0)
    hh, hw = np.float32(1) - lh, np.float32(1) - lw
    t, b, l, r = h0 >= 0, h1 <= Hl - 1, w0 >= 0, w1 <= Wl - 1
    off = lambda ok, hh_, ww_: base + (hh_ * Wl + ww_) * row_stride if ok else -1
    return ((off(t and l, h0, w0), off(t and r, h0, w1), off(b and l, h1, w0), off(b and r, h1, w1)),
            (hh * hw, hh * lw, lh * hw, lh * lw), lh, lw)


def emulate(value, shapes, lsi, loc, attw, grad_out):
    B, Nv, H, C = value.shape
    _, Nq, _, L, P, _ = loc.shape
    assert C == K_CH
    LP = L * P
    v = value.reshape(-1).astype(np.float32)
    locf, wf, gof = loc.reshape(-1), attw.reshape(-1), grad_out.reshape(-1)
    gv = np.zeros_like(v)
    gl = np.full(locf.shape, np.nan, np.float32)
    gw = np.full(wf.shape, np.nan, np.float32)
    n_bq = B * Nq
    n_hg = (H + K_PHEADS - 1) // K_PHEADS
    nblocks = ((n_bq + 1) // 2) * n_hg
    row_stride = H * K_CH
    requests = merged = 0
    for blk in range(nblocks):
        pair, h0 = blk // n_hg, (blk % n_hg) * K_PHEADS
        nh = min(K_PHEADS, H - h0)
        bq0 = pair * 2
        nhalf = 2 if bq0 + 1 < n_bq else 1
        s_loc = np.zeros((2 * K_PHEADS, LP * 2), np.float32)
        s_w = np.zeros((2 * K_PHEADS, LP), np.float32)
        for half in range(nhalf):                                   # staging
            first = (bq0 + half) * H + h0
            s_loc.reshape(-1)[half * K_PHEADS * LP * 2: half * K_PHEADS * LP * 2 + nh * LP * 2] = \
                locf[first * LP * 2: first * LP * 2 + nh * LP * 2]
            s_w.reshape(-1)[half * K_PHEADS * LP: half * K_PHEADS * LP + nh * LP] = wf[first * LP: first * LP + nh * LP]
        for wv in range(nh):                                        # one wave per head
            st = []
            for half in range(2):
                live = half < nhalf
                bq = bq0 + (half if live else 0)
                h = h0 + wv
                b = bq // Nq
                item = bq * H + h
                gbase = b * Nv * row_stride + h * K_CH
                go = gof[item * K_CH: item * K_CH + K_CH] if live else np.zeros(K_CH, np.float32)
                st.append(dict(live=live, gbase=gbase, go=go, slot=half * K_PHEADS + wv))
            for l in range(L):
                Hl, Wl = int(shapes[l, 0]), int(shapes[l, 1])
                base = int(lsi[l]) * row_stride
                for p in range(P):
                    o = [[-1] * 4, [-1] * 4]
                    val = [[np.zeros(K_CH, np.float32)] * 4, [np.zeros(K_CH, np.float32)] * 4]
                    g = [None, None]
                    for half in range(2):
                        s = st[half]
                        gx = gy = gw_ = np.zeros(K_CH, np.float32)
                        if s["live"]:
                            x = s_loc[s["slot"], (l * P + p) * 2] * np.float32(Wl) - np.float32(0.5)
                            y = s_loc[s["slot"], (l * P + p) * 2 + 1] * np.float32(Hl) - np.float32(0.5)
                            w = s_w[s["slot"], l * P + p]
                            if y > -1 and x > -1 and y < Hl and x < Wl:
                                offs, cw, lh, lw = corners(x, y, Hl, Wl, base, row_stride)
                                ch = np.arange(K_CH)
                                d = [v[s["gbase"] + oo + ch] * s["go"] if oo >= 0 else np.zeros(K_CH, np.float32) for oo in offs]
                                hh, hw = np.float32(1) - lh, np.float32(1) - lw
                                gw_ = cw[0] * d[0] + cw[1] * d[1] + cw[2] * d[2] + cw[3] * d[3]
                                gx = w * Wl * (-hh * d[0] + hh * d[1] - lh * d[2] + lh * d[3])
                                gy = w * Hl * (-hw * d[0] - lw * d[1] + hw * d[2] + lw * d[3])
                                wg = w * s["go"]
                                for i in range(4):
                                    if offs[i] >= 0:
                                        o[half][i] = s["gbase"] + offs[i]
                                        val[half][i] = cw[i] * wg
                        g[half] = (gx.sum(), gy.sum(), gw_.sum())
                    for half in range(2):                           # exchange + merge
                        po, pv = o[1 - half], val[1 - half]
                        for i in range(4):
                            total = val[half][i].copy()
                            issue = o[half][i] >= 0
                            if half == 0:
                                for j in range(4):
                                    if issue and po[j] == o[half][i]:
                                        total = total + pv[j]; merged += 1
                            else:
                                for j in range(4):
                                    if po[j] >= 0 and po[j] == o[half][i]:
                                        issue = False
                            if issue:
                                gv[o[half][i]: o[half][i] + K_CH] += total
                                requests += 1
                    for half in range(2):
                        if st[half]["live"]:
                            s = st[half]
                            s_loc[s["slot"], (l * P + p) * 2] = g[half][0]
                            s_loc[s["slot"], (l * P + p) * 2 + 1] = g[half][1]
                            s_w[s["slot"], l * P + p] = g[half][2]
        for half in range(nhalf):                                   # write-back
            first = (bq0 + half) * H + h0
            gl[first * LP * 2: first * LP * 2 + nh * LP * 2] = \
                s_loc.reshape(-1)[half * K_PHEADS * LP * 2: half * K_PHEADS * LP * 2 + nh * LP * 2]
            gw[first * LP: first * LP + nh * LP] = s_w.reshape(-1)[half * K_PHEADS * LP: half * K_PHEADS * LP + nh * LP]
    return gv.reshape(value.shape), gl.reshape(loc.shape), gw.reshape(attw.shape), requests, merged


def check(B, shapes, Nq, H, P, coherent, seed=0):
    value, sh, loc, w = M.make_case(seed, B, shapes, Nq, P=P)
    if H != value.shape[2]:
        value, loc, w = value[:, :, :H].contiguous(), loc[:, :, :H].contiguous(), w[:, :, :H].contiguous()
    if coherent:                                                    # neighbouring queries sample next to each other
        g = torch.Generator().manual_seed(1)
        base = torch.stack([torch.arange(Nq) * 0.013 + 0.2, torch.full((Nq,), 0.4)], -1)
        off = (torch.rand(1, 1, H, len(shapes), P, 2, generator=g) - 0.5) * 0.05
        loc = (base[None, :, None, None, None, :] + off).expand_as(loc).contiguous()
    lsi = M.level_start_index(shapes)
    vg = value.clone().requires_grad_(True); lg = loc.clone().requires_grad_(True); wg = w.clone().requires_grad_(True)
    out = M.msda_gather(vg.double(), sh, lg.double(), wg.double())
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    out.backward(go.double())
    gv, gl, gw, req, merged = emulate(value.numpy(), sh.numpy(), lsi.numpy(), loc.numpy(), w.numpy(),
                                      go.float().numpy().reshape(B, Nq, H * 32))
    plain = int(B * Nq * H * len(shapes) * P * 4)
    for name, a, b in (("grad_value", gv, vg.grad), ("grad_loc", gl, lg.grad), ("grad_w", gw, wg.grad)):
        b = b.numpy()
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
        assert np.isfinite(a).all() and err < 2e-4, (name, err)
    print(f"B={B} Nq={Nq} H={H} L={len(shapes)} P={P} coherent={coherent}: ok, atomic requests {req} "
          f"(plain kernel <= {plain}), merged corner pairs {merged}")


if __name__ == "__main__":
    check(2, [(6, 7), (3, 4)], 5, 8, 2, coherent=False)      # odd query count: a lone last half
    check(1, [(9, 11)], 8, 8, 4, coherent=True)
    check(2, [(8, 10), (4, 5), (2, 3)], 6, 2, 3, coherent=True)   # 2 heads: partial head group
    check(1, [(5, 5)], 1, 6, 2, coherent=False)              # one query, 6 heads: two head groups, no partner
