"""csrc/conv3x3_mfma.hip restated lane by lane in numpy: the slab staging (pixels [p0 - W - 1, ...), zero outside the
image), the packed weights, the 32x32x2 MFMA operand / accumulator mapping, the left / right padding flags and the
XCD-aware tile order -- checked against torch's conv2d on the CPU, so that an indexing slip never costs a GPU call."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

TP, OB = 128, 32


def pack_weights(w, cout):
    C = w.shape[1]
    wt = np.zeros(C * 9 * OB, np.float32)
    for e in range(wt.size):
        o, tap, c = e & (OB - 1), (e >> 5) % 9, e // (9 * OB)
        if o < cout:
            wt[e] = w[o, c].reshape(9)[tap]
    return wt


def mfma_32x32x2(a, b, acc):
    """a[lane], b[lane]: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; acc[lane][r] = D[row][lane & 31]
    with row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)"""
    A = np.zeros((32, 2), np.float64); B = np.zeros((2, 32), np.float64)
    for lane in range(64):
        A[lane & 31, lane >> 5] = a[lane]
        B[lane >> 5, lane & 31] = b[lane]
    D = A @ B
    for lane in range(64):
        for r in range(16):
            acc[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]


def emulate(x, w, bias, CK=8):
    CPW = CK // 4
    N, C, H, W = x.shape
    cout = w.shape[0]
    HW = H * W
    tiles = (HW + TP - 1) // TP
    total = tiles * N
    per_xcd = (total + 7) // 8
    iters = (TP + 2 * W + 2 + 63) // 64
    SW = 64 * iters + 32
    wt = pack_weights(w, cout)
    out = np.full((N, cout, HW), np.nan, np.float32)
    xf = x.reshape(N, C, HW)
    seen = set()
    for block in range(per_xcd * 8):
        t = (block & 7) * per_xcd + (block >> 3)
        if t >= total:
            continue
        assert t not in seen
        seen.add(t)
        n, p0 = t // tiles, (t % tiles) * TP
        q0 = p0 - W - 1
        acc = np.zeros((4, 64, 16), np.float64)
        for c0 in range(0, C, CK):
            slab = np.full(CK * SW, np.nan, np.float32)
            for wave in range(4):
                for u in range(CPW):
                    for i in range(iters):
                        for lane in range(64):
                            q = q0 + lane + 64 * i
                            slab[(CPW * wave + u) * SW + lane + 64 * i] = xf[n, c0 + CPW * wave + u, q] if 0 <= q < HW else 0.0
            wl = wt[c0 * 9 * OB:(c0 + CK) * 9 * OB]
            for wave in range(4):
                for cp in range(CK // 2):
                    for ky in range(3):
                        for kx in range(3):
                            tap = ky * 3 + kx
                            a = np.zeros(64); b = np.zeros(64)
                            for lane in range(64):
                                l31, h = lane & 31, lane >> 5
                                p = p0 + 32 * wave + l31
                                col = p % W
                                a[lane] = wl[h * 9 * OB + l31 + (2 * cp * 9 + tap) * OB]
                                v = slab[h * SW + 32 * wave + l31 + 2 * cp * SW + ky * W + kx]
                                if kx == 0 and col == 0:
                                    v = 0.0
                                if kx == 2 and col == W - 1:
                                    v = 0.0
                                b[lane] = v
                            assert not np.isnan(b).any()
                            mfma_32x32x2(a, b, acc[wave])
        for wave in range(4):
            for lane in range(64):
                p = p0 + 32 * wave + (lane & 31)
                if p >= HW:
                    continue
                for r in range(16):
                    o = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                    if o < cout:
                        assert np.isnan(out[n, o, p])
                        out[n, o, p] = acc[wave, lane, r] + (bias[o] if bias is not None else 0.0)
    assert len(seen) == total
    return out.reshape(N, cout, H, W)


@pytest.mark.parametrize("N,C,H,W,cout,with_bias", [(1, 8, 5, 7, 27, True), (2, 16, 9, 50, 27, True), (1, 8, 3, 100, 32, False),
                                                     (3, 8, 2, 1, 5, True), (1, 8, 12, 11, 1, True)])
def test_lane_level_restatement_matches_conv2d(N, C, H, W, cout, with_bias):
    rng = np.random.default_rng(N * 100 + C + H + W)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = rng.standard_normal((cout, C, 3, 3)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if with_bias else None
    got = emulate(x, w, b)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                   torch.from_numpy(b).double() if b is not None else None, padding=1).numpy()
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_bank_layout_of_the_operand_reads():
    """a wave's A / B reads: 32 consecutive dwords per half, the second half 32 banks away (SW and 9 * 32 are == 32 mod 64)"""
    for W in (1, 50, 100, 191):
        iters = (TP + 2 * W + 2 + 63) // 64
        SW = 64 * iters + 32
        assert SW % 64 == 32 and iters <= 8
        assert 32 * 3 + 31 + 2 * W + 2 < 64 * iters            # the farthest slab column a lane reads was staged
    assert (9 * OB) % 64 == 32
