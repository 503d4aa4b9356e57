"""CPU: vectorised ray-error metrics vs golden values from the reference's utils/eval_utils.py
(tests/golden/make_eval_golden.py).  The spherical nearest-neighbour goes through an injected CPU
chamfer so that the test needs no GPU; the GPU variant is tests/test_eval_utils_gpu.py."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import chamfer as C
from vidar_amd.plugin.utils import eval_utils as E

G = np.load(Path(__file__).parent / "golden" / "eval_ray_errors.npz")


class CpuChamfer(torch.nn.Module):
    def forward(self, src, tgt, bidirectional=False, reverse=False, reduction="mean"):
        idx, d = C.knn_points_idx(tgt.numpy(), src.numpy())
        return torch.from_numpy(d[..., 0]).sum(1).mean(), (torch.from_numpy(d[..., 0]), torch.from_numpy(idx[..., 0]))


@pytest.mark.parametrize("name", ["origin_inside", "origin_outside"])
def test_clamp_matches_reference(name):
    o, p, inv = E.clamp(G[f"{name}_gt"], G[f"{name}_origin"], return_invalid_mask=True)
    assert np.array_equal(inv, G[f"{name}_invalid"])
    np.testing.assert_allclose(p[~inv], G[f"{name}_clamp_p"][~inv], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(o[~inv], G[f"{name}_clamp_o"][~inv], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("name", ["origin_inside", "origin_outside"])
def test_ray_errors_match_reference(name):
    l1, ar = E.compute_ray_errors(G[f"{name}_pred"].copy(), G[f"{name}_gt"].copy(), G[f"{name}_origin"].copy(),
                                  torch.device("cpu"), chamfer=CpuChamfer())
    np.testing.assert_allclose(l1, G[f"{name}_l1"], rtol=1e-9)
    np.testing.assert_allclose(ar, G[f"{name}_absrel"], rtol=1e-9)
