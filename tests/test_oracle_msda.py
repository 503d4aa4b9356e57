"""CPU: the two independent MSDA oracle formulations agree (forward and autograd backward)."""
import pytest
import torch

from oracle import msda as M

CASES = [
    (1, [(8, 8)], 50, 4),                       # single level (TSA / prediction shape family)
    (2, [(12, 20), (6, 10), (3, 5), (2, 3)], 77, 8),   # 4 FPN levels x 8 points (SCA family)
    (1, [(1, 1)], 3, 1),
]


@pytest.mark.parametrize("B,shapes,Nq,P", CASES)
def test_gather_equals_grid_sample(B, shapes, Nq, P):
    value, sh, loc, w = M.make_case(0, B, shapes, Nq, P=P, dtype=torch.float64)
    value.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    a = M.msda_gather(value, sh, loc, w)
    ga = torch.autograd.grad(a.square().sum(), [value, loc, w])
    b = M.msda_grid_sample(value, sh, loc, w)
    gb = torch.autograd.grad(b.square().sum(), [value, loc, w])
    torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-10)
    for x, y in zip(ga, gb):
        torch.testing.assert_close(x, y, rtol=1e-8, atol=1e-8)


def test_weights_linearity_and_padding():
    value, sh, loc, w = M.make_case(1, 1, [(5, 7)], 20, P=4, dtype=torch.float64)
    out = M.msda_gather(value, sh, loc, w)
    torch.testing.assert_close(M.msda_gather(value, sh, loc, 2 * w), 2 * out)
    far = torch.full_like(loc, 3.0)              # every sample outside -> zeros
    assert float(M.msda_gather(value, sh, far, w).abs().sum()) == 0.0
