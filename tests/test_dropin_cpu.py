"""CPU: the reference's OWN wrapper files, imported UNCHANGED from /root/reference, run on top of the
shims once `vidar_amd.dropin.install()` answers their loader calls
  torch.utils.cpp_extension.load("dvxlr"/"dvxlr_v2", ...)   (utils/e2e_predictor_utils.py:86-90, :118-121)
  mmcv.utils.ext_loader.load_ext('_ext', [...])              (modules/multi_scale_deformable_attn_function.py:11-12)
and the shims are importable under the reference's top-level names (dvr, dvxlr, dvxlr_v2, chamferdist).
No GPU here, so the shims' entry points are routed to the CPU oracle (same signatures): what this proves is
the SURFACE -- names, argument order, keyword `im2col_step`, list returns, caller-allocated gradient
buffers.  The same call patterns run on the HIP kernels in tests/test_dropin_gpu.py.
Skipped where /root/reference is absent (the GPU box)."""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

REF = Path("/root/reference/projects/mmdet3d_plugin/bevformer")
GOLD = Path(__file__).parent / "golden"
needs_ref = pytest.mark.skipif(not REF.exists(), reason="/root/reference not present")


def _load_unchanged(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture
def dropin():
    from vidar_amd import dropin as D
    stubbed = []
    if "mmcv" not in sys.modules:                         # mmcv is not installable here: loader namespace only
        for n in ("mmcv", "mmcv.utils", "mmcv.utils.ext_loader"):
            sys.modules[n] = types.ModuleType(n); stubbed.append(n)
        sys.modules["mmcv"].utils = sys.modules["mmcv.utils"]
        sys.modules["mmcv.utils"].ext_loader = sys.modules["mmcv.utils.ext_loader"]
        sys.modules["mmcv.utils.ext_loader"].load_ext = lambda *a, **k: (_ for _ in ()).throw(ImportError("no mmcv._ext"))
    D.install()
    yield D
    D.uninstall()
    for n in stubbed:
        sys.modules.pop(n, None)


def test_top_level_names_resolve_to_the_shims(dropin):
    import chamferdist
    import dvr
    import dvxlr
    import dvxlr_v2
    from chamferdist import _C
    for mod, names in ((dvr, ("init", "render", "render_forward")),
                       (dvxlr, ("init", "render", "get_grad_sigma")),
                       (dvxlr_v2, ("render_v2", "get_grad_sigma_v2")),
                       (chamferdist, ("ChamferDistance", "knn_points")),
                       (_C, ("knn_points_idx", "knn_points_backward", "knn_check_version"))):
        for n in names:
            assert callable(getattr(mod, n)), (mod.__name__, n)
    assert dvxlr.__name__ == "vidar_amd.third_lib.dvxlr"
    with pytest.raises(RuntimeError):
        dropin.load("some_other_extension", sources=[])


def test_uninstall_restores_the_loaders():
    import torch.utils.cpp_extension as E
    from vidar_amd import dropin as D
    before = E.load
    D.install(); assert E.load is D.load
    D.uninstall(); assert E.load is before and "dvxlr" not in sys.modules


@needs_ref
def test_reference_e2e_predictor_utils_runs_unchanged_on_the_shims(dropin):
    from test_e2e_utils_golden_cpu import dvxlr_on_oracle
    sys.path.insert(0, str(GOLD))
    from make_e2e_utils_golden import ray_case
    ref = _load_unchanged("ref_e2e_predictor_utils", REF / "utils/e2e_predictor_utils.py")
    assert ref.dvxlr.__name__ == "vidar_amd.third_lib.dvxlr" and ref.dvxlr_v2.__name__ == "vidar_amd.third_lib.dvxlr_v2"
    gold = np.load(GOLD / "e2e_utils.npz")
    sigma, origin, points, tindex = ray_case()
    with dvxlr_on_oracle():
        s = sigma.clone().requires_grad_(True)
        p, g = ref.DifferentiableVoxelRendering(s, origin, points, tindex)
        w = torch.from_numpy(gold["l1_w"])
        (p * w).sum().backward()
        np.testing.assert_allclose(p.detach().numpy(), gold["l1_pred"], rtol=2e-5, atol=1e-4)
        np.testing.assert_array_equal(g.detach().numpy(), gold["l1_gt"])
        np.testing.assert_allclose(s.grad.numpy(), gold["l1_grad"], rtol=1e-4, atol=1e-5 * np.abs(gold["l1_grad"]).max())
        s2 = sigma.clone().requires_grad_(True)
        reg = torch.from_numpy(gold["l2_reg"]).clone().requires_grad_(True)
        p2, g2, rp, ind = ref.DifferentiableVoxelRenderingV2(s2, origin, points, tindex, reg)
        ((p2 * w).sum() + (rp * torch.from_numpy(gold["l2_wr"]) * (ind.detach() >= 0)).sum()).backward()
        np.testing.assert_array_equal(ind.detach().numpy(), gold["l2_indicator"])
        np.testing.assert_allclose(s2.grad.numpy(), gold["l2_grad"], rtol=1e-4, atol=1e-5 * np.abs(gold["l2_grad"]).max())
        np.testing.assert_allclose(reg.grad.numpy(), gold["l2_grad_reg"], rtol=1e-5, atol=1e-6)


@needs_ref
def test_reference_msda_function_runs_unchanged_on_the_shim(dropin, monkeypatch):
    from oracle import msda as M
    from vidar_amd.third_lib import mmcv_ext
    ref = _load_unchanged("ref_msda_function", REF / "modules/multi_scale_deformable_attn_function.py")
    assert ref.ext_module is mmcv_ext
    seen = {}

    def fwd(value, shapes, lsi, loc, w, im2col_step=64):
        seen["fwd_step"] = im2col_step
        return M.msda_gather(value, shapes, loc, w)

    def bwd(value, shapes, lsi, loc, w, grad_out, grad_value, grad_loc, grad_w, im2col_step=64):
        seen["bwd_step"] = im2col_step
        assert not grad_value.any() and not grad_loc.any() and not grad_w.any()     # caller pre-zeroes
        with torch.enable_grad():                    # we are inside a once_differentiable backward
            v, l_, w_ = (t.detach().double().requires_grad_(True) for t in (value, loc, w))
            g = torch.autograd.grad((M.msda_gather(v, shapes, l_, w_) * grad_out.double()).sum(), [v, l_, w_])
        grad_value.copy_(g[0]); grad_loc.copy_(g[1]); grad_w.copy_(g[2])

    monkeypatch.setattr(mmcv_ext, "ms_deform_attn_forward", fwd)
    monkeypatch.setattr(mmcv_ext, "ms_deform_attn_backward", bwd)
    value, sh, loc, w = M.make_case(0, 2, [(6, 5), (3, 3)], 7, P=4)
    lsi = M.level_start_index([(6, 5), (3, 3)])
    v, l_, w_ = value.requires_grad_(True), loc.requires_grad_(True), w.requires_grad_(True)
    out = ref.MultiScaleDeformableAttnFunction_fp32.apply(v, sh, lsi, l_, w_, 64)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    got = torch.autograd.grad((out * gout).sum(), [v, l_, w_])
    assert seen == {"fwd_step": 64, "bwd_step": 64}
    v2, l2, w2 = (t.detach().double().requires_grad_(True) for t in (value, loc, w))
    want = torch.autograd.grad((M.msda_gather(v2, sh, l2, w2) * gout.double()).sum(), [v2, l2, w2])
    for a, b in zip(got, want):
        torch.testing.assert_close(a.double(), b, rtol=1e-5, atol=1e-6)
