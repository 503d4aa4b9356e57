"""CPU: the generic (D, K) nearest-neighbour program of `chamferdist._C` (the reference dispatches these to templated
kernels, knn.cu:266-295; ViDAR itself only calls D = 3, K = 1) against the reference's OWN knn_cpu.cpp build
(oracle/_ref): indices, distances and both gradients, with ragged lengths, ties and K above the cloud size.  The
program is device-agnostic torch code; here its two functions run on host tensors, the GPU test drives the public
entry points."""
import numpy as np
import pytest
import torch

from vidar_amd.third_lib.chamferdist import _C


def _clouds(seed, N, P1, P2, D, ties):
    rng = np.random.default_rng(seed)
    if ties:                                   # a coarse lattice: many exactly equal distances
        a = rng.integers(-3, 4, (N, P1, D)).astype(np.float32)
        b = rng.integers(-3, 4, (N, P2, D)).astype(np.float32)
    else:
        a = rng.standard_normal((N, P1, D)).astype(np.float32)
        b = rng.standard_normal((N, P2, D)).astype(np.float32)
    l1 = rng.integers(max(P1 // 2, 0), P1 + 1, N).astype(np.int64)
    l2 = rng.integers(0, P2 + 1, N).astype(np.int64)
    return tuple(torch.from_numpy(x) for x in (a, b, l1, l2))


@pytest.mark.parametrize("D,K", [(3, 2), (3, 5), (2, 1), (5, 3), (8, 4), (1, 2), (3, 40)])
@pytest.mark.parametrize("ties", [False, True])
def test_generic_knn_matches_the_reference_build(D, K, ties, ref_modules):
    ref = ref_modules("ref_chamferdist_C")
    a, b, l1, l2 = _clouds(D * 100 + K, 2, 57, 33, D, ties)
    l2[0] = 33; l2[1] = min(int(l2[1]), 3)                       # one full cloud, one shorter than most K
    ri, rd = ref.knn_points_idx(a, b, l1, l2, K, -1)
    oi, od = _C._generic_knn_idx(a, b, l1, l2, K)
    assert torch.equal(oi, ri)
    assert torch.equal(od, rd)                                   # same fp32 operation order as the CPU build
    g = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 57, K)).astype(np.float32))
    r1, r2 = ref.knn_points_backward(a, b, l1, l2, ri, g)
    o1, o2 = _C._generic_knn_backward(a, b, l1, l2, oi, g)
    torch.testing.assert_close(o1, r1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o2, r2, rtol=1e-5, atol=1e-5)


def test_generic_knn_chunks_and_empty_inputs(monkeypatch, ref_modules):
    ref = ref_modules("ref_chamferdist_C")
    a, b, l1, l2 = _clouds(0, 1, 200, 90, 3, False)
    ri, rd = ref.knn_points_idx(a, b, l1, l2, 3, -1)
    monkeypatch.setattr(_C, "_GENERIC_CHUNK_ELEMS", 90 * 7)      # 7 rows of p1 per chunk
    oi, od = _C._generic_knn_idx(a, b, l1, l2, 3)
    assert torch.equal(oi, ri) and torch.equal(od, rd)
    e = torch.zeros(1, 0, 3)
    i0, d0 = _C._generic_knn_idx(a, e, l1, torch.zeros(1, dtype=torch.int64), 2)
    assert i0.shape == (1, 200, 2) and not i0.any() and not d0.any()
    i1, d1 = _C._generic_knn_idx(e, b, torch.zeros(1, dtype=torch.int64), l2, 2)
    assert i1.shape == (1, 0, 2)


def test_version_table_follows_the_reference():
    # KnnCheckVersion (knn.cu:269-280)
    assert _C.knn_check_version(0, 100, 100)
    assert _C.knn_check_version(1, 32, 100) and not _C.knn_check_version(1, 33, 1)
    assert _C.knn_check_version(2, 8, 32) and not _C.knn_check_version(2, 8, 33) and not _C.knn_check_version(2, 9, 1)
    assert _C.knn_check_version(3, 8, 4) and not _C.knn_check_version(3, 8, 5)
    assert not _C.knn_check_version(4, 3, 1) and not _C.knn_check_version(-1, 3, 1)


def test_public_entry_points_still_refuse_host_tensors():
    a, b, l1, l2 = _clouds(0, 1, 8, 8, 3, False)
    with pytest.raises(RuntimeError):
        _C.knn_points_idx(a, b, l1, l2, 2, -1)
