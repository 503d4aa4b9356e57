"""GPU parity: HIP MSDA forward/backward vs the CPU oracle (fp64 gather formulation) and vs golden vectors
made by the reference's own in-tree copy of this arithmetic (`dcnv3_core_pytorch`, tests/golden/make_msda_golden.py).
Tolerance: fp32 accumulation of <=32 products per output -> rtol 1e-4 / atol 1e-5 (the op-level tolerance;
the oracle is pinned against the reference's DCNv3 code, oracle/msda.py, tests/test_oracle_msda.py)."""
import pytest
import torch

from oracle import msda as M

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["head_major", "banded"])
def item_order(request):
    """every case under both workgroup -> item orders of the gather kernels (vidar_msda_set_item_order): results must not
    depend on which 32 (batch, query, head) items a workgroup owns"""
    from vidar_amd._lib import lib
    prev = lib().vidar_msda_set_item_order(1 if request.param == "head_major" else 0)
    yield request.param
    lib().vidar_msda_set_item_order(prev)

CASES = [
    ("tsa_small", 2, [(20, 20)], 400, 4),
    ("sca_small", 3, [(12, 20), (6, 10), (3, 5), (2, 3)], 333, 8),
    ("pred_small", 1, [(16, 12)], 192, 4),
    ("one", 1, [(1, 1)], 1, 1),
    ("ragged_items", 1, [(9, 7)], 5, 4),          # 40 items: last workgroup partially filled
    ("bev_self_attn", 2, [(24, 24)], 576, 4),         # Nv == Nq square grid (tile edge 8 divides the grid)
    ("bev_ragged_tiles", 1, [(20, 20)], 400, 4),      # grid not a multiple of the 8x8 destination tile
    ("sca_mid", 2, [(30, 50), (15, 25), (8, 13), (4, 7)], 1000, 8),  # multi-level, several chunks per tile
    ("two_heads", 2, [(12, 20), (6, 10)], 144, 8, 2),  # the golden models' width: embed 64 = 2 heads
    ("hot_tile", 1, [(3, 3)], 5000, 8),                # > 1024 samples per destination tile -> several chunks
    ("max_samples_per_item", 1, [(10, 12), (5, 6), (3, 3), (2, 2)], 70, 16),   # L*P = 64: 66.5 KB of record LDS (opt-in > 64 KiB)
]
SCATTER = {"atomic": False, "binned": True}


def run(B, shapes, Nq, P, H=8, seed=0, binned=None):
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    F32 = F.MultiScaleDeformableAttnFunction_fp32
    value, sh, loc, w = M.make_case(seed, B, shapes, Nq, H=H, P=P)
    v64, l64, w64 = value.double().requires_grad_(True), loc.double().requires_grad_(True), \
        w.double().requires_grad_(True)
    ref = M.msda_gather(v64, sh, l64, w64)
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(seed + 1), dtype=torch.float64)
    gref = torch.autograd.grad((ref * gout).sum(), [v64, l64, w64])
    dv, dl, dw = value.cuda().requires_grad_(True), loc.cuda().requires_grad_(True), \
        w.cuda().requires_grad_(True)
    out = F32.apply(dv, sh.cuda(), M.level_start_index(shapes).cuda(), dl, dw, 64)
    if binned is None:
        got = torch.autograd.grad((out * gout.float().cuda()).sum(), [dv, dl, dw])
    else:        # the scatter strategy is the caller's choice (workspace or not, include/vidar_hip.h)
        got = F._msda_backward(dv.detach(), sh.cuda(), M.level_start_index(shapes).cuda(), dl.detach(),
                               dw.detach(), gout.float().cuda(), binned=binned)
    return ref, gref, out, got


@pytest.mark.parametrize("scatter", list(SCATTER))
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_msda_fwd_bwd(case, scatter):
    name, B, shapes, Nq, P = case[:5]
    H = case[5] if len(case) > 5 else 8
    ref, gref, out, got = run(B, shapes, Nq, P, H=H, binned=SCATTER[scatter])
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-5)
    for g, r, nm in zip(got, gref, ["grad_value", "grad_loc", "grad_w"]):
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(g.cpu().double(), r, rtol=2e-4, atol=2e-5 * scale, msg=lambda m: nm + m)


@pytest.mark.parametrize("scatter", list(SCATTER))
@pytest.mark.parametrize("name", ["tsa_L1_P4", "sca_L4_P8"])
def test_msda_matches_reference_dcnv3_golden(name, scatter):
    """HIP forward + the three gradients against vectors the REFERENCE produced (fp64 `dcnv3_core_pytorch` per level,
    summed over levels, autograd gradients) on fp32-representable value / weights / grad_out; the fp64 locations are
    rounded to fp32 for the device, hence the location-rounding term in the tolerances (|d out / d pixel| ~ 3)."""
    import numpy as np
    from pathlib import Path
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    d = np.load(Path(__file__).parent / "golden" / f"msda_{name}.npz")
    shapes = [(int(a), int(b)) for a, b in d["shapes"]]
    sh = torch.from_numpy(d["shapes"]).cuda()
    lsi = M.level_start_index(shapes).cuda()
    value = torch.from_numpy(d["value"]).cuda()
    loc = torch.from_numpy(d["loc"]).float().cuda()
    w = torch.from_numpy(d["w"]).cuda()
    gout = torch.from_numpy(d["gout"]).cuda()
    out = F.MultiScaleDeformableAttnFunction_fp32.apply(value, sh, lsi, loc, w, 64)
    torch.testing.assert_close(out.cpu().double(), torch.from_numpy(d["out"]), rtol=1e-4, atol=5e-5)
    got = F._msda_backward(value, sh, lsi, loc, w, gout, binned=SCATTER[scatter])
    for g, nm in zip(got, ["grad_value", "grad_loc", "grad_w"]):
        r = torch.from_numpy(d[nm])
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(g.cpu().double(), r, rtol=2e-4, atol=5e-5 * scale, msg=lambda m: nm + m)


def test_empty_queries():
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import multi_scale_deformable_attn
    value, sh, loc, w = M.make_case(0, 1, [(4, 4)], 0, P=4)
    out = multi_scale_deformable_attn(value.cuda(), sh.cuda(), M.level_start_index([(4, 4)]).cuda(),
                                      loc.cuda(), w.cuda())
    assert out.shape == (1, 0, 256)


def test_full_size_tsa_linearity():
    """TSA shape of BASELINE config 1 (B=2, 200x200, 4 points): linear in value and in weights."""
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import multi_scale_deformable_attn as f
    shapes = [(200, 200)]
    value, sh, loc, w = M.make_case(3, 2, shapes, 40000, P=4)
    value, sh, loc, w = value.cuda(), sh.cuda(), loc.cuda(), w.cuda()
    lsi = M.level_start_index(shapes).cuda()
    a = f(value, sh, lsi, loc, w)
    torch.testing.assert_close(f(2 * value, sh, lsi, loc, w), 2 * a, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(f(value, sh, lsi, loc, 0.5 * w), 0.5 * a, rtol=1e-5, atol=1e-5)
    ones = torch.ones_like(value)
    inside = loc.clamp(0.05, 0.95)              # constant field + interior samples -> sum of weights
    c = f(ones, sh, lsi, inside, w)
    torch.testing.assert_close(c, w.sum((3, 4)).repeat_interleave(32, -1).view_as(c), rtol=1e-5, atol=1e-5)


import functools


@functools.lru_cache(maxsize=None)
def _full_case(name):
    B, shapes, Nq, P = FULL[name]
    value, sh, loc, w = M.make_case(11, B, shapes, Nq, P=P)
    gout = torch.randn(B, Nq, 256, generator=torch.Generator().manual_seed(12))
    return value, sh, loc, w, gout, _oracle_fwd_bwd_per_batch(value, sh, loc, w, gout)


def _oracle_fwd_bwd_per_batch(value, sh, loc, w, gout):
    """fp64 grid_sample formulation, one batch element at a time (memory)."""
    outs, gv, gl, gw = [], [], [], []
    for b in range(value.shape[0]):
        v, l_, w_ = (t[b:b + 1].double().requires_grad_(True) for t in (value, loc, w))
        o = M.msda_grid_sample(v, sh, l_, w_)
        g = torch.autograd.grad((o * gout[b:b + 1].double()).sum(), [v, l_, w_])
        outs.append(o.detach()); gv.append(g[0]); gl.append(g[1]); gw.append(g[2])
    return torch.cat(outs), torch.cat(gv), torch.cat(gl), torch.cat(gw)


FULL = {
    # BASELINE config 1 shapes: TSA B=2 200x200 P=4; SCA B=6 cams, 4 FPN levels of the
    # 928x1600 image (Nv=30 825), max_len 10^4 visible queries, P=8; Prediction B=1 200x200 P=4
    "tsa_200x200": (2, [(200, 200)], 40000, 4),
    "sca_6cam_fpn": (6, [(116, 200), (58, 100), (29, 50), (15, 25)], 10000, 8),
    "pred_200x200": (1, [(200, 200)], 40000, 4),
}


@pytest.mark.parametrize("scatter", list(SCATTER))
@pytest.mark.parametrize("name", list(FULL))
def test_full_size_matches_oracle(name, scatter):
    """HIP forward + backward against the fp64 oracle at the BASELINE shapes (both scatter strategies)."""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    B, shapes, Nq, P = FULL[name]
    value, sh, loc, w, gout, (ref, rv, rl, rw) = _full_case(name)
    lsi = M.level_start_index(shapes).cuda()
    dv, dl, dw = value.cuda(), loc.cuda(), w.cuda()
    out = F._msda_forward(dv, sh.cuda(), lsi, dl, dw)
    # tolerance at 200-pixel-wide levels: the fp32 pixel coordinate loc*W-0.5 carries ulp(200)/2 = 7.6e-6
    # of rounding that the fp64 oracle does not have; times |d value / d pixel| ~ 3 (unit-variance value)
    # -> a few 1e-5 absolute on outputs of magnitude ~1 (the fp32 reference kernel has the same rounding)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-4)
    got = [g.cpu().double() for g in F._msda_backward(dv, sh.cuda(), lsi, dl, dw, gout.cuda(), binned=SCATTER[scatter])]
    # d/d location of a bilinear sample jumps at pixel boundaries (and at the -1 / size borders): a sample
    # whose exact pixel coordinate lies within fp32 rounding of an integer may fall into the neighbouring
    # cell in fp32 (a few of the 5-15 M samples do).  Those samples are excluded from the grad_loc check;
    # the forward, grad_value and grad_w are continuous there and are compared everywhere.
    wh = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64).view(1, 1, 1, len(shapes), 1, 2)
    pixel = loc.double() * wh - 0.5
    # only samples that can contribute (inside the zero-padding border, with the same margin) need the exemption:
    # outside, both sides have a zero gradient.  The exempted share is asserted, not just bounded: a window of
    # +-1e-4 px around the integers on two axes is 4e-4 of the live samples -- a kernel that is wrong near cell
    # borders cannot hide in it.
    live = ((pixel > -1 - 1e-4) & (pixel < wh + 1e-4)).all(-1, keepdim=True)
    kink = (((pixel - pixel.round()).abs() < 1e-4).any(-1, keepdim=True) & live).expand_as(loc)
    share = float(kink[..., 0].double().sum() / live.double().sum())
    print(f"{name}: {int(kink[..., 0].sum())} of {int(live.sum())} live samples exempted from the grad_loc check ({share:.2e})")
    assert 2e-4 < share < 8e-4, share
    got[1] = torch.where(kink, rl, got[1])
    for g, r, nm in zip(got, (rv, rl, rw), ["grad_value", "grad_loc", "grad_w"]):
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(g, r, rtol=2e-4, atol=1e-4 * scale, msg=lambda m: nm + m)


def test_binned_workspace_contract():
    """workspace too small -> invalid argument; unsupported level count -> 0 bytes (caller falls back)."""
    import ctypes
    from vidar_amd._lib import lib, ptr, stream_of
    f = lib().vidar_msda_bwd_workspace_bytes
    f.restype = ctypes.c_size_t
    assert f(2, 400, 8, 400, 1, 4) > 0
    assert f(1, 400, 8, 400, 17, 1) == 0
    value, sh, loc, w = M.make_case(0, 2, [(20, 20)], 400, P=4)
    a = [t.cuda() for t in (value, sh, M.level_start_index([(20, 20)]), loc, w)]
    go = torch.zeros(2, 400, 256, device="cuda")
    gv, gl, gw = torch.empty_like(a[0]), torch.empty_like(a[3]), torch.empty_like(a[4])
    ws = torch.empty(8, dtype=torch.int64, device="cuda")
    rc = lib().vidar_msda_bwd_f32(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), ptr(go), ptr(gv), ptr(gl),
                                  ptr(gw), 2, 400, 8, 32, 400, 1, 4, ptr(ws), ctypes.c_size_t(64), stream_of(go))
    assert rc == -22


FUSED = [  # name, bs, Qn, shapes, Nq, P, mode, R
    ("tsa_like", 2, 2, [(20, 20)], 400, 4, 0, 1),
    ("pred_cross_2_levels", 1, 1, [(12, 12), (12, 12)], 144, 4, 0, 2),
    ("sca_like", 3, 1, [(12, 20), (6, 10), (3, 5), (2, 3)], 333, 8, 1, 4),
    ("sca_two_heads", 2, 1, [(12, 20), (6, 10)], 50, 8, 1, 4),
    ("tsa_binned_size", 1, 2, [(100, 100)], 10000, 4, 0, 1),           # 640 k samples -> binned backward
]


@pytest.mark.parametrize("case", FUSED, ids=[c[0] for c in FUSED])
def test_fused_operand_preparation_matches_the_reference_tensor_program(case):
    """vidar_msda_fused_{fwd,bwd}: raw sampling_offsets / attention_weights Linear outputs in, gradients w.r.t.
    them out -- against the reference's own tensor program (softmax, /normalizer, + reference points, queue
    permutes) in fp64 on the CPU followed by the gather oracle."""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    name, bs, Qn, shapes, Nq, P, mode, R = case
    H = 2 if name == "sca_two_heads" else 8
    L = len(shapes)
    g = torch.Generator().manual_seed(len(name))
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(bs * Qn, Nv, H, 32, generator=g)
    off_raw = torch.randn(bs, Nq, H * Qn * L * P * 2, generator=g) * 3.0            # pixels
    logit_raw = torch.randn(bs, Nq, H * Qn * L * P, generator=g)
    ref = torch.rand(bs * Qn, Nq, R, 2, generator=g) * 1.1 - 0.05
    sh = torch.tensor(shapes, dtype=torch.int64)
    lsi = M.level_start_index(shapes)
    gout = torch.randn(bs * Qn, Nq, H * 32, generator=g)
    v64, o64, l64 = (t.double().requires_grad_(True) for t in (value, off_raw, logit_raw))
    loc, w = F.compose_operands(o64, l64, ref.double(), sh, H, Qn, L, P, mode)
    want = M.msda_gather(v64, sh, loc, w)
    gwant = torch.autograd.grad((want * gout.double()).sum(), [v64, o64, l64])
    dv, do, dl = (t.cuda().requires_grad_(True) for t in (value, off_raw, logit_raw))
    out = F.fused_deform_attn(dv, sh.cuda(), lsi.cuda(), do, dl, ref.cuda(), Qn, L, P, mode)
    torch.testing.assert_close(out.detach().cpu().double(), want.detach(), rtol=1e-4, atol=2e-5)
    got = [t.cpu().double() for t in torch.autograd.grad((out * gout.cuda()).sum(), [dv, do, dl])]
    # samples within fp32 rounding of a pixel boundary: d/d offset is one-sided there (see the full-size test)
    wh = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64).view(1, 1, 1, L, 1, 2)
    pixel = loc.detach() * wh - 0.5
    live = ((pixel > -1 - 1e-4) & (pixel < wh + 1e-4)).all(-1, keepdim=True)
    kink = (((pixel - pixel.round()).abs() < 1e-4).any(-1, keepdim=True) & live).expand_as(loc)   # [bs*Qn,Nq,H,L,P,2]
    # small cases: the expected share is 4e-4 of the live samples, a handful in absolute terms
    assert float(kink[..., 0].double().sum()) <= max(8.0, 2e-3 * float(live.double().sum()))
    if mode == 0:        # back to the raw layout [bs, Nq, H, Qn, L, P, 2]
        kink = kink.view(bs, Qn, Nq, H, L, P, 2).permute(0, 2, 3, 1, 4, 5, 6)
    kink = kink.reshape(got[1].shape)
    got[1] = torch.where(kink, gwant[1], got[1])
    for a, b, nm in zip(got, gwant, ["grad_value", "grad_off_raw", "grad_logit_raw"]):
        scale = max(1.0, float(b.abs().max()))
        torch.testing.assert_close(a, b, rtol=3e-4, atol=3e-5 * scale, msg=lambda m: nm + m)


@pytest.mark.parametrize("scatter", list(SCATTER))
def test_samples_outside_the_level_and_nan_locations_contribute_nothing(scatter):
    """every sample off the level (or NaN): zero output, zero gradients, nothing binned -- both strategies"""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    shapes = [(10, 12), (5, 6)]
    value, sh, loc, w = M.make_case(2, 2, shapes, 300, P=4)
    loc = loc + 5.0                                   # far outside [0, 1]
    loc[0, :7] = float("nan")
    lsi = M.level_start_index(shapes).cuda()
    out = F._msda_forward(value.cuda(), sh.cuda(), lsi, loc.cuda(), w.cuda())
    assert float(out.abs().max()) == 0.0
    gv, gl, gw = F._msda_backward(value.cuda(), sh.cuda(), lsi, loc.cuda(), w.cuda(),
                                  torch.randn(2, 300, 256, device="cuda"), binned=SCATTER[scatter])
    assert float(gv.abs().max()) == 0.0 and float(gl.abs().max()) == 0.0 and float(gw.abs().max()) == 0.0


def test_one_hot_destination_pixel_many_chunks():
    """all 40 000 x 8 x 4 samples land on the same 2x2 pixels: one destination tile, > 1000 chunks flushing onto
    the same window lines (the binned path's worst case) still equals the atomic scatter"""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    shapes = [(16, 16)]
    value, sh, loc, w = M.make_case(3, 1, shapes, 40000, P=4)
    loc = torch.full_like(loc, 0.5) + (torch.rand(loc.shape, generator=torch.Generator().manual_seed(1)) - 0.5) * 0.05
    lsi = M.level_start_index(shapes).cuda()
    go = torch.randn(1, 40000, 256, generator=torch.Generator().manual_seed(2)).cuda()
    a = F._msda_backward(value.cuda(), sh.cuda(), lsi, loc.cuda(), w.cuda(), go, binned=True)
    b = F._msda_backward(value.cuda(), sh.cuda(), lsi, loc.cuda(), w.cuda(), go, binned=False)
    for u, v in zip(a, b):
        torch.testing.assert_close(u, v, rtol=2e-3, atol=2e-3 * max(1.0, float(v.abs().max())))


@pytest.mark.parametrize("scatter", list(SCATTER))
def test_rows_with_nan_locations_between_live_rows(scatter):
    """queries whose every sample location is NaN mixed with live ones inside the same waves: a NaN location is
    "outside" for every kernel -- those rows are 0 with zero gradients, the live rows equal the run without them bit for bit"""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    value, sh, loc, w = M.make_case(5, 3, shapes, 333, P=8)
    dead = torch.zeros(3, 333, dtype=torch.bool)
    dead[:, ::3] = True; dead[1, 100:200] = True; dead[2, -40:] = True
    loc_nan = loc.clone(); loc_nan[dead] = float("nan")
    lsi = M.level_start_index(shapes).cuda()
    go = torch.randn(3, 333, 256, generator=torch.Generator().manual_seed(6))
    go[dead] = 0.0                                   # what the scatter-back's backward hands to padded slots
    args = (value.cuda(), sh.cuda(), lsi)
    out_ref = F._msda_forward(*args, loc.cuda(), w.cuda())
    out = F._msda_forward(*args, loc_nan.cuda(), w.cuda())
    d = dead.cuda()
    assert float(out[d].abs().max()) == 0.0
    assert torch.equal(out[~d], out_ref[~d])
    gv, gl, gw = F._msda_backward(*args, loc_nan.cuda(), w.cuda(), go.cuda(), binned=SCATTER[scatter])
    gv0, gl0, gw0 = F._msda_backward(*args, loc.cuda(), w.cuda(), go.cuda(), binned=SCATTER[scatter])
    assert float(gl[d].abs().max()) == 0.0 and float(gw[d].abs().max()) == 0.0
    assert torch.equal(gl[~d], gl0[~d]) and torch.equal(gw[~d], gw0[~d])
    torch.testing.assert_close(gv, gv0, rtol=1e-4, atol=1e-5 * max(1.0, float(gv0.abs().max())))   # zero grad_out rows add 0


@pytest.mark.parametrize("scatter", list(SCATTER))
def test_value_tensor_past_4_gib_is_split_over_batch_elements(scatter):
    """`value` of 4.4 GB (9 x 480 000 pixels x 8 heads x 32 channels): one launch addresses at most 4 GiB with its 32-bit byte
    offsets, so the entry points split the call over batch elements (8 + 1 here) -- the result must equal the per-element calls"""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    B, shapes, Nq, P = 9, [(600, 800)], 64, 4
    g = torch.Generator(device="cuda").manual_seed(3)
    value = torch.randn(B, 480000, 8, 32, device="cuda", generator=g)
    assert value.numel() * 4 > (1 << 32)
    sh = torch.tensor(shapes, dtype=torch.int64, device="cuda")
    lsi = M.level_start_index(shapes).cuda()
    loc = torch.rand(B, Nq, 8, 1, P, 2, device="cuda", generator=g)
    w = torch.softmax(torch.randn(B, Nq, 8, P, device="cuda", generator=g), -1).view(B, Nq, 8, 1, P)
    go = torch.randn(B, Nq, 256, device="cuda", generator=g)
    out = F._msda_forward(value, sh, lsi, loc, w)
    gv, gl, gw = F._msda_backward(value, sh, lsi, loc, w, go, binned=SCATTER[scatter])
    for b in (0, 4, 8):
        sl = slice(b, b + 1)
        assert torch.equal(out[sl], F._msda_forward(value[sl], sh, lsi, loc[sl], w[sl]))
        gv1, gl1, gw1 = F._msda_backward(value[sl], sh, lsi, loc[sl], w[sl], go[sl], binned=SCATTER[scatter])
        assert torch.equal(gl[sl], gl1) and torch.equal(gw[sl], gw1)
        torch.testing.assert_close(gv[sl], gv1, rtol=1e-5, atol=1e-6)      # atomics: summation order


@pytest.mark.parametrize("bs,Nq,shapes,P", [(1, 576, [(24, 24)], 4), (2, 333, [(12, 20)], 4), (1, 10000, [(100, 100)], 4),
                                            (2, 70, [(9, 7), (5, 4)], 3)])
def test_merged_queue_mean_equals_the_unmerged_op_and_its_mean(bs, Nq, shapes, P):
    """merge_queue (TemporalSelfAttention: the mean over the two BEV queue entries inside the gather): output and the
    three gradients equal the unmerged fused op followed by `.view(bs, Qn, Nq, C).mean(1)` -- both the one-launch atomic
    backward (small cases) and the binned one (the 10 000-query case), and a launch split over batch rows"""
    from vidar_amd.plugin.modules import multi_scale_deformable_attn_function as F
    Qn, H, L = 2, 8, len(shapes)
    g = torch.Generator().manual_seed(Nq)
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(bs * Qn, Nv, H, 32, generator=g).cuda()
    off_raw = (torch.randn(bs, Nq, H * Qn * L * P * 2, generator=g) * 3.0).cuda()
    logit_raw = torch.randn(bs, Nq, H * Qn * L * P, generator=g).cuda()
    ref = (torch.rand(bs * Qn, Nq, L, 2, generator=g) * 1.1 - 0.05).cuda()
    sh = torch.tensor(shapes, dtype=torch.int64).cuda()
    lsi = M.level_start_index(shapes).cuda()
    gout = torch.randn(bs, Nq, H * 32, generator=g).cuda()
    res = []
    for merge in (True, False):
        dv, do, dl = (t.clone().requires_grad_(True) for t in (value, off_raw, logit_raw))
        out = F.fused_deform_attn(dv, sh, lsi, do, dl, ref, Qn, L, P, 0, merge_queue=merge)
        if not merge:
            assert out.shape == (bs * Qn, Nq, H * 32)
            out = out.view(bs, Qn, Nq, H * 32).mean(1)
        assert out.shape == (bs, Nq, H * 32)
        res.append((out.detach(), torch.autograd.grad((out * gout).sum(), [dv, do, dl])))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    for a, b, nm in zip(res[0][1], res[1][1], ["grad_value", "grad_off_raw", "grad_logit_raw"]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(b.abs().max())), msg=lambda m: nm + m)
