"""The thread program of csrc/affine_act.hip's stem_pool_kernel (frozen BN + ReLU + 3x3/2 max-pool in one pass),
restated with numpy loops over (row pair r, column pair t) exactly as the kernel indexes, against
max_pool2d(relu(x*s+b)): checks the window / border mapping without a GPU (the GPU test compares the kernel itself,
bit for bit, tests/test_dcn_gpu.py::test_fused_stem_pool_matches_two_kernel_path)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def thread_program(x, s, b):
    N, C, H, W = x.shape
    assert W % 4 == 0
    Ho, Wo = (H - 1) // 2 + 1, W // 2
    y = np.full((N, C, Ho, Wo), np.nan, np.float32)
    f = lambda v, c: np.maximum(v * s[c] + b[c], np.float32(0))
    for n in range(N):
        for c in range(C):
            for r in range((Ho + 1) // 2):
                for t in range(W // 4):
                    m = np.full(4, -np.inf, np.float32)          # m0a, m0b, m1a, m1b
                    for i in range(5):
                        row = 4 * r - 1 + i
                        if row < 0 or row >= H:
                            continue
                        v = f(x[n, c, row, 4 * t:4 * t + 4], c)
                        left = f(x[n, c, row, 4 * t - 1], c) if t > 0 else np.float32(-np.inf)
                        a, c2 = max(left, v[0], v[1]), max(v[1], v[2], v[3])
                        if i <= 2:
                            m[0], m[1] = max(m[0], a), max(m[1], c2)
                        if i >= 2:
                            m[2], m[3] = max(m[2], a), max(m[3], c2)
                    y[n, c, 2 * r, 2 * t:2 * t + 2] = m[:2]
                    if 2 * r + 1 < Ho:
                        y[n, c, 2 * r + 1, 2 * t:2 * t + 2] = m[2:]
    return y


@pytest.mark.parametrize("shape", [(1, 2, 8, 8), (2, 3, 7, 12), (1, 1, 2, 4), (1, 2, 1, 4), (1, 1, 9, 16), (1, 2, 6, 20)])
def test_thread_program_equals_bn_relu_maxpool(shape):
    rng = np.random.default_rng(0)
    x = rng.normal(size=shape).astype(np.float32)
    s = rng.uniform(0.5, 1.5, shape[1]).astype(np.float32)
    b = rng.normal(size=shape[1]).astype(np.float32)
    ref = F.max_pool2d(F.relu(torch.from_numpy(x) * torch.from_numpy(s)[None, :, None, None]
                              + torch.from_numpy(b)[None, :, None, None]), 3, stride=2, padding=1).numpy()
    out = thread_program(x, s, b)
    assert out.shape == ref.shape and not np.isnan(out).any()     # every output written
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)       # numpy's a*s+b is not fused: last-bit slack only
