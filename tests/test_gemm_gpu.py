"""csrc/gemm_mfma.hip through the C ABI (vidar_gemm_f32) against fp64 products.

VIDAR_GEMM_F32 must be an fp32 GEMM (error of the class of torch's own fp32 product); VIDAR_GEMM_BF16X3 must be at least
four times tighter than a product of TF32-rounded operands -- the arithmetic the reference executes these layers in
(README.md:96: torch 1.10.1+cu111 defaults allow_tf32 = True; tools/train.py:141-144) -- on the same operands, at the
BASELINE shapes of the attention value projection (spatial_cross_attention.py:333-340: [6*30825, 256] x [256, 256]).
Tolerances are written where they are used."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vidar_amd import gemm as G  # noqa: E402

DEV = "cuda"


def tf32_round(t):
    """round-to-nearest-even to 10 stored mantissa bits (what an A100 tensor core reads of an fp32 operand)"""
    i = t.contiguous().view(torch.int32)
    i = (i + 0xFFF + ((i >> 13) & 1)) & ~0x1FFF
    return i.view(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(DEV)


def normwise(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


def logical(t, layout, rows, cols):
    """the tensor that stores a logical [rows, cols] operand in `layout` (0: cols contiguous, 1: rows contiguous)"""
    return t if layout == 0 else t.t()


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
@pytest.mark.parametrize("a_layout", [0, 1])
@pytest.mark.parametrize("b_layout", [0, 1])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (130, 70, 45), (257, 129, 100), (64, 300, 16), (1, 1, 1), (37, 5, 263)])
def test_layouts_and_ragged_shapes(precision, a_layout, b_layout, M, N, K):
    # A logical [M,K]: K-major storage [M,K]; MN-major storage [K,M].  B logical [K,N]: K-major storage [N,K]
    a_store = rnd(M, K, seed=1) if a_layout == 0 else rnd(K, M, seed=1)
    b_store = rnd(N, K, seed=2) if b_layout == 0 else rnd(K, N, seed=2)
    A = a_store if a_layout == 0 else a_store.t()
    B = b_store.t() if b_layout == 0 else b_store
    ref = A.double() @ B.double()
    # C is a window of a larger NaN-filled buffer: nothing outside the window may be written (rows past M are cut by
    # the store descriptor's range check, columns past N by the kernel's own test)
    big = torch.full((M + 40, N + 8), float("nan"), device=DEV)
    C = big[:M, :N]
    G.gemm_raw(a_store, a_store.stride(0), a_layout, b_store, b_store.stride(0), b_layout, C, big.stride(0), M, N, K,
               precision=precision)
    assert torch.isfinite(C).all()
    assert torch.isnan(big[M:]).all() and torch.isnan(big[:, N:]).all(), "wrote outside the [M, N] window"
    # fp32 mode: K <= 263 products of |x| <= 1 -> 2e-6 of the largest entry; bf16x3: 3 * 2^-16 per product -> 1e-4 bound
    tol = 2e-6 if precision == G.F32 else 6e-5
    assert normwise(C, ref) <= tol, normwise(C, ref)


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
def test_identity_times_asymmetric_matrix(precision):
    """A = I, B asymmetric: a transposed accumulator write-out cannot pass; exact in both modes for bf16-exact B"""
    n = 160
    A = torch.eye(n, device=DEV)
    B = (torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) % 251)       # integers < 256: exact in bf16
    for bl in (0, 1):
        b_store = B.t().contiguous() if bl == 0 else B
        C = torch.empty(n, n, device=DEV)
        G.gemm_raw(A, n, 0, b_store, n, bl, C, n, n, n, n, precision=precision)
        assert torch.equal(C, B)


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
@pytest.mark.parametrize("vec_axis", [0, 1])
def test_epilogue_scale_shift_residual_relu(precision, vec_axis):
    M, N, K = 200, 136, 64
    A, Bs = rnd(M, K, seed=3), rnd(N, K, seed=4)
    nv = N if vec_axis == 0 else M
    scale, shift, res = rnd(nv, seed=5) + 1.5, rnd(nv, seed=6), rnd(M, N, seed=7)
    acc = A.double() @ Bs.double().t()
    bc = (lambda v: v.double()[None, :]) if vec_axis == 0 else (lambda v: v.double()[:, None])
    ref = torch.relu(acc * bc(scale) + bc(shift) + res.double())
    big = torch.full((M + 3, N), float("nan"), device=DEV)     # the residual's descriptor ends at row M: no read past it
    C = big[:M]
    G.gemm_raw(A, K, 0, Bs, K, 0, C, N, M, N, K, scale=scale, shift=shift, vec_axis=vec_axis, residual=res, ldr=N,
               relu=True, precision=precision)
    assert torch.isnan(big[M:]).all()
    assert normwise(C, ref) <= (2e-6 if precision == G.F32 else 6e-5)
    assert float((C == 0).float().mean()) > 0.2 and float(C.min()) >= 0.0          # the ReLU was applied


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
def test_batched_and_reduced(precision):
    Bn, Co, Ci, HW = 3, 96, 80, 1450                     # HW = 1450: rows start 8-byte aligned only
    w, x = rnd(Co, Ci, seed=8), rnd(Bn, Ci, HW, seed=9)
    y = G.conv_forward(w, x, precision=precision)
    ref = torch.einsum("oc,bcp->bop", w.double(), x.double())
    tol = 2e-6 if precision == G.F32 else 6e-5
    assert normwise(y, ref) <= tol
    g = rnd(Bn, Co, HW, seed=10)
    gx = G.conv_grad_input(w, g, precision)
    assert normwise(gx, torch.einsum("oc,bop->bcp", w.double(), g.double())) <= tol
    gw = G.conv_grad_weight(g, x, precision)                # batch-reduced, split over HW: slabs summed in order
    assert normwise(gw, torch.einsum("bop,bcp->oc", g.double(), x.double())) <= 2 * tol
    gw2 = G.conv_grad_weight(g, x, precision)
    assert torch.equal(gw, gw2), "the slab reduction must be deterministic"


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
@pytest.mark.parametrize("rows,N,K", [(5000, 37, 5), (4099, 3, 3), (70000, 130, 66)])
def test_split_k_weight_gradient_with_ragged_sizes(precision, rows, N, K):
    """g^T x over many rows (split over the rows, slabs summed in a fixed order), output sizes that are not multiples
    of anything: the slab reduction's float4 body and its scalar tail"""
    g, x = rnd(rows, N, seed=21), rnd(rows, K, seed=22)
    gw = G.linear_grad_weight(g, x, precision)
    ref = g.double().t() @ x.double()
    assert normwise(gw, ref) <= (3e-6 if precision == G.F32 else 8e-5)
    assert torch.equal(gw, G.linear_grad_weight(g, x, precision))
    # the bias gradient rides along (a_rowsum): same weight gradient bit for bit, g.sum(0) in a fixed order
    gw2, gb = G.linear_grad_weight(g, x, precision, with_bias=True)
    assert torch.equal(gw2, gw)
    assert normwise(gb, g.double().sum(0)) <= 2e-6          # exact fp32 adds in both modes (no split on this path)
    gw3, gb3 = G.linear_grad_weight(g, x, precision, with_bias=True)
    assert torch.equal(gb, gb3) and torch.equal(gw3, gw), "weight and bias gradient must be deterministic"


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
def test_bias_gradient_rides_along_at_the_baseline_shapes(precision):
    """grad_out [184 950, 256] (SpatialCrossAttention value projection) and [40 000, 512] (FFN): the row sums written next
    to the weight gradient equal grad_out.sum(0), including a row-sliced (strided) grad_out and a single-tile product"""
    for rows, N, K in ((184950, 256, 256), (40000, 512, 256), (300, 128, 256)):
        g, x = rnd(rows, N, seed=31), rnd(rows, K, seed=32)
        gw, gb = G.linear_grad_weight(g, x, precision, with_bias=True)
        assert normwise(gb, g.double().sum(0)) <= 3e-6
        assert normwise(gw, g.double().t() @ x.double()) <= (3e-6 if precision == G.F32 else 8e-5)
    wide = rnd(5000, 300, seed=33)
    g = wide[:, 20:150]                                       # leading dimension 300 > 130 columns
    gw, gb = G.linear_grad_weight(g, rnd(5000, 64, seed=34), precision, with_bias=True)
    assert normwise(gb, g.double().sum(0)) <= 3e-6


@pytest.mark.parametrize("M,K,N", [(184950, 256, 256), (80000, 256, 256), (40000, 512, 128)])
def test_bf16x3_at_least_four_times_tighter_than_tf32_at_baseline_shapes(M, K, N):
    """the acceptance criterion of the bf16x3 path: max error vs the fp64 product <= 1/4 of the error of the same
    product with TF32-rounded operands (fp64 accumulate -- the most favourable TF32 result) on the same operands"""
    x, w, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.1), rnd(N, seed=13)
    ref = torch.empty(M, N, dtype=torch.float64, device=DEV)
    tf = torch.empty(M, N, dtype=torch.float64, device=DEV)
    xt, wt = tf32_round(x).double(), tf32_round(w).double()
    wd = w.double()
    for s in range(0, M, 32768):                           # fp64 products in slices (memory)
        ref[s:s + 32768] = x[s:s + 32768].double() @ wd.t() + b.double()
        tf[s:s + 32768] = xt[s:s + 32768] @ wt.t() + b.double()
    err_tf32 = float((tf - ref).abs().max())
    y3 = G.linear_forward(x, w, b, False, G.BF16X3)
    err_x3 = float((y3.double() - ref).abs().max())
    y1 = G.linear_forward(x, w, b, False, G.F32)
    err_f32 = float((y1.double() - ref).abs().max())
    lib = torch.addmm(b, x, w.t())
    err_lib = float((lib.double() - ref).abs().max())
    print(f"[{M},{K}]x[{K},{N}] max abs error: tf32 {err_tf32:.3e}  bf16x3 {err_x3:.3e}  f32-mfma {err_f32:.3e}  "
          f"library fp32 {err_lib:.3e}")
    assert err_x3 <= 0.25 * err_tf32, (err_x3, err_tf32)
    assert err_f32 <= 4 * err_lib + 1e-7                   # same arithmetic class as the library's fp32 kernel


@pytest.mark.parametrize("m", ["f32", "bf16x3"])
@pytest.mark.parametrize("relu", [False, True])
def test_linear_autograd_matches_torch(m, relu):
    x = rnd(2, 300, 256, seed=14).requires_grad_(True)
    w = rnd(128, 256, seed=15, scale=0.2).requires_grad_(True)
    b = rnd(128, seed=16).requires_grad_(True)
    gy = rnd(2, 300, 128, seed=17)
    y = G.linear(x, w, b, relu=relu, m=m)
    y.backward(gy)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = torch.nn.functional.linear(xd, wd, bd)
    if relu:                                   # the kernel's own ReLU mask: an entry within rounding of 0 may flip
        yd = yd * (y.detach() > 0).double()
    yd.backward(gy.double())
    tol = 3e-6 if m == "f32" else 6e-5
    assert normwise(y.detach(), yd.detach()) <= tol
    for a, r in zip(got, (xd.grad, wd.grad, bd.grad)):
        assert normwise(a, r) <= 2 * tol


def test_bad_arguments_are_rejected():
    a = rnd(8, 8)
    with pytest.raises(ValueError):
        G.gemm_raw(a, 4, 0, a, 8, 0, a.clone(), 8, 8, 8, 8)              # lda < K
    with pytest.raises(ValueError):
        G.gemm_raw(a, 8, 2, a, 8, 0, a.clone(), 8, 8, 8, 8)              # unknown layout
    with pytest.raises(RuntimeError):
        G.linear_forward(a.cpu(), a.cpu())


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
def test_wave_specialised_variants_are_bit_identical(variant, precision):
    """vidar_gemm_set_variant(1 / 2): the wave-specialised workgroups (4 staging + 4 matrix waves, LDS double buffer) run
    the same arithmetic in the same tile order as the default kernel -- every layout pair, ragged shapes, a batch, and the
    full epilogue, compared bit for bit with variant 0"""
    from vidar_amd._lib import lib
    cases = [(130, 70, 45, 1), (257, 129, 100, 1), (128, 256, 64, 3), (37, 5, 263, 2), (1, 1, 1, 1)]

    def run():
        outs = []
        for M, N, K, batch in cases:
            for al in (0, 1):
                for bl in (0, 1):
                    a = rnd(batch, M, K, seed=M) if al == 0 else rnd(batch, K, M, seed=M)
                    b = rnd(batch, N, K, seed=N) if bl == 0 else rnd(batch, K, N, seed=N)
                    scale, shift, res = rnd(N, seed=3), rnd(N, seed=4), rnd(batch, M, N, seed=5)
                    C = torch.empty(batch, M, N, device=DEV)
                    G.gemm_raw(a, a.stride(1), al, b, b.stride(1), bl, C, N, M, N, K, batch=batch, sA=a.stride(0),
                               sB=b.stride(0), sC=M * N, scale=scale, shift=shift, vec_axis=0, residual=res, ldr=N,
                               sR=M * N, relu=True, precision=precision)
                    outs.append(C)
        return outs
    base = run()
    prev = lib().vidar_gemm_set_variant(variant)
    try:
        got = run()
    finally:
        lib().vidar_gemm_set_variant(prev)
    assert prev == 0
    for a, b in zip(got, base):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision", [G.F32, G.BF16X3])
@pytest.mark.parametrize("M,N,K", [(1000, 67, 40), (4097, 130, 33), (513, 6, 256)])
def test_rowsum_with_ragged_rows(precision, M, N, K):
    """weight + bias gradient of a Linear whose output width N is NOT a multiple of 4, with lda == N: the float4 loads of
    the MN-major grad_out straddle the row end and read the next k-row's first elements -- those components must be
    dropped by the row guard of the row-sum store (csrc/gemm_mfma.hip, add_rowsum), never summed"""
    g2 = rnd(M, N, seed=7)
    g2[5, N - 1] = 1.0e4                        # a large value next to the ragged edge: a leaked lane would show
    x2 = rnd(M, K, seed=8)
    gw, gb = G.linear_grad_weight(g2, x2, precision, with_bias=True)
    ref_b = g2.double().sum(0)
    ref_w = g2.double().t() @ x2.double()
    assert torch.isfinite(gb).all() and torch.isfinite(gw).all()
    tol = 3e-6 if precision == G.F32 else 1e-4
    assert float((gb.double() - ref_b).abs().max() / ref_b.abs().max()) <= 2e-5
    assert normwise(gw, ref_w) <= tol * (M ** 0.5)
