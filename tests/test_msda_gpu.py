"""GPU parity: HIP MSDA forward/backward vs the CPU oracle (fp64 gather formulation).
Tolerance: fp32 accumulation of <=32 products per output -> rtol 1e-4 / atol 1e-5 (stated here as
the op-level tolerance; the oracle itself is 'parity unpinned', see oracle/msda.py)."""
import pytest
import torch

from oracle import msda as M

pytestmark = pytest.mark.gpu

CASES = [
    ("tsa_small", 2, [(20, 20)], 400, 4),
    ("sca_small", 3, [(12, 20), (6, 10), (3, 5), (2, 3)], 333, 8),
    ("pred_small", 1, [(16, 12)], 192, 4),
    ("one", 1, [(1, 1)], 1, 1),
    ("ragged_items", 1, [(9, 7)], 5, 4),          # 40 items: last workgroup partially filled
    ("bev_self_attn_tiled", 2, [(24, 24)], 576, 4),   # Nv == Nq square grid -> 8x8 tiled write-combining path
    ("bev_ragged_tiles", 1, [(20, 20)], 400, 4),      # grid not a multiple of the 8x8 tile
    ("sca_wc", 2, [(30, 50), (15, 25), (8, 13), (4, 7)], 1000, 8),  # multi-level write-combining path
]


def run(B, shapes, Nq, P, seed=0):
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import \
        MultiScaleDeformableAttnFunction_fp32 as F32
    value, sh, loc, w = M.make_case(seed, B, shapes, Nq, P=P)
    v64, l64, w64 = value.double().requires_grad_(True), loc.double().requires_grad_(True), \
        w.double().requires_grad_(True)
    ref = M.msda_gather(v64, sh, l64, w64)
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(seed + 1), dtype=torch.float64)
    gref = torch.autograd.grad((ref * gout).sum(), [v64, l64, w64])
    dv, dl, dw = value.cuda().requires_grad_(True), loc.cuda().requires_grad_(True), \
        w.cuda().requires_grad_(True)
    out = F32.apply(dv, sh.cuda(), M.level_start_index(shapes).cuda(), dl, dw, 64)
    got = torch.autograd.grad((out * gout.float().cuda()).sum(), [dv, dl, dw])
    return ref, gref, out, got


@pytest.mark.parametrize("name,B,shapes,Nq,P", CASES)
def test_msda_fwd_bwd(name, B, shapes, Nq, P):
    ref, gref, out, got = run(B, shapes, Nq, P)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-5)
    for g, r, nm in zip(got, gref, ["grad_value", "grad_loc", "grad_w"]):
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(g.cpu().double(), r, rtol=2e-4, atol=2e-5 * scale, msg=lambda m: nm + m)


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
