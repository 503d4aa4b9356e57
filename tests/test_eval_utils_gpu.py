"""GPU: ray-error metrics with the HIP nearest-neighbour vs the reference's golden values."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent / "golden" / "eval_ray_errors.npz")


@pytest.mark.parametrize("name", ["origin_inside", "origin_outside"])
def test_ray_errors_on_gpu(name):
    from vidar_amd.plugin.utils import eval_utils as E
    l1, ar = E.compute_ray_errors(G[f"{name}_pred"].copy(), G[f"{name}_gt"].copy(), G[f"{name}_origin"].copy(),
                                  torch.device("cuda"))
    # fp32 spherical coordinates: a near-tie may pick another neighbour; metrics agree to 1e-3
    np.testing.assert_allclose(l1, G[f"{name}_l1"], rtol=1e-3)
    np.testing.assert_allclose(ar, G[f"{name}_absrel"], rtol=1e-3)
