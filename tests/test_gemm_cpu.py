"""CPU: the numerics model of VIDAR_GEMM_BF16X3 (csrc/gemm_mfma.hip) restated in numpy, and the host logic of
vidar_amd/gemm.py.  The model is what the device code does to an operand while it stages it: hi = bf16_rne(x),
lo = bf16_rne(x - hi), product = lo*hi + hi*lo + hi*hi accumulated in fp32."""
import numpy as np
import pytest
import torch


def bf16_rne(x):
    i = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    i = (i + 0x7FFF + ((i >> 16) & 1)) & 0xFFFF0000
    return i.astype(np.uint32).view(np.float32)


def tf32_rne(x):
    i = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    i = (i + 0xFFF + ((i >> 13) & 1)) & 0xFFFFE000
    return i.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_rne(x)
    lo = bf16_rne((x - hi).astype(np.float32))
    return hi, lo


def test_split_carries_sixteen_significand_bits():
    rng = np.random.default_rng(0)
    x = (rng.uniform(-1, 1, 1 << 16) * np.exp2(rng.integers(-20, 20, 1 << 16))).astype(np.float32)
    hi, lo = split(x)
    assert np.all((x - hi).astype(np.float32).astype(np.float64) == x.astype(np.float64) - hi.astype(np.float64)), \
        "x - hi must be exact in fp32"
    rel = np.abs(x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)) / np.abs(x)
    assert rel.max() <= 2.0 ** -16
    assert (np.abs(x.astype(np.float64) - tf32_rne(x)) / np.abs(x)).max() > 2.0 ** -12     # TF32: 11 bits


@pytest.mark.parametrize("M,K,N", [(512, 256, 256), (256, 1024, 64)])
def test_three_term_product_is_at_least_four_times_tighter_than_tf32(M, K, N):
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    b = (rng.uniform(-1, 1, (K, N)) * 0.1).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    ah, al = split(a); bh, bl = split(b)
    d = lambda x: x.astype(np.float64)
    x3 = (d(al) @ d(bh) + d(ah) @ d(bl) + d(ah) @ d(bh)).astype(np.float32)      # exact products, one final rounding
    tf = d(tf32_rne(a)) @ d(tf32_rne(b))
    err3, errt = np.abs(x3 - ref).max(), np.abs(tf - ref).max()
    assert err3 <= 0.25 * errt, (err3, errt)


def test_mode_switch_and_no_cpu_path():
    from vidar_amd import gemm as G
    assert G.mode() in ("lib", "auto", "f32", "bf16x3")
    assert not G.own_kernels("auto") and not G.own_kernels("lib") and G.own_kernels("f32") and G.own_kernels("bf16x3")
    prev = G.mode()
    with G.use("bf16x3"):
        assert G.mode() == "bf16x3" and G.precision_of() == G.BF16X3
        with G.use("f32"):
            assert G.precision_of() == G.F32
        assert G.mode() == "bf16x3"
    assert G.mode() == prev
    with pytest.raises(ValueError):
        G.set_mode("tf32")
    with pytest.raises(RuntimeError):
        G.linear_forward(torch.zeros(4, 8), torch.zeros(2, 8))


def test_split_plan_fills_the_chip_and_is_bounded():
    import ctypes
    from vidar_amd._lib import lib
    L = lib()
    L.vidar_gemm_workspace_bytes.restype = ctypes.c_size_t
    # grad_weight of the value projection: [256, 184950] x [184950, 256] -> 4 tiles: many splits, >= 4 k-steps each
    s = L.vidar_gemm_splits(256, 256, 184950, 1, 1, 1)
    assert 64 <= s <= 256 and (184950 + s - 1) // s >= 4 * 32
    # s slabs of [M, N] + s partial rows of [M] for the optional row sums of A (the bias gradient that rides along)
    assert L.vidar_gemm_workspace_bytes(256, 256, 184950, 1, 1, 1) == s * 256 * 256 * 4          # the product alone
    assert L.vidar_gemm_workspace_bytes(256, 256, 184950, 1, 1, 2) == s * (256 * 256 + 256) * 4  # ... with A's row sums
    assert L.vidar_gemm_splits(256, 256, 184950, 1, 1, 0) == 1 and L.vidar_gemm_workspace_bytes(256, 256, 64, 1, 0, 0) == 0
    # a single-slab reduced product still answers for one slab + one row: the row sums force the slab path
    assert L.vidar_gemm_splits(128, 128, 32, 1, 1, 1) == 1
    # a single slab is written straight to C: no scratch unless the row sums ride along
    assert L.vidar_gemm_workspace_bytes(128, 128, 32, 1, 1, 1) == 0
    assert L.vidar_gemm_workspace_bytes(128, 128, 32, 1, 1, 2) == (128 * 128 + 128) * 4
