"""Per-kernel timing of the HIP ops at BASELINE sizes (HIP events on torch's current stream, which
is the stream every op launches on).  Prints one JSON line per op: ms, algorithmic GB/s."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def timeit(fn, warm=3, it=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def report(name, ms, nbytes=None, **kw):
    d = {"op": name, "ms": round(ms, 4)}
    if nbytes:
        d["alg_MB"] = round(nbytes / 1e6, 2); d["GBps"] = round(nbytes / ms / 1e6, 1)
        d["frac_8TBps"] = round(nbytes / ms / 1e6 / 8000, 4)
    d.update(kw)
    print(json.dumps(d), flush=True)


PAD_MODE_DEFAULT = 1


def bench_dvr(which):
    buf = torch.empty(2478800000 // 4, device="cuda")
    ms = timeit(lambda: buf.zero_())
    report("device fill 2.48 GB (write ceiling)", ms, buf.numel() * 4)
    src = torch.empty_like(buf)
    ms = timeit(lambda: buf.copy_(src))
    report("device copy 2.48 GB (read+write bytes)", ms, buf.numel() * 8)
    del buf, src
    from vidar_amd.synthetic import ray_set
    from vidar_amd.third_lib import dvr, dvxlr, dvxlr_v2
    from vidar_amd._lib import lib
    t = lambda a: torch.from_numpy(a).cuda()
    # A/B of the launch variants (results are identical, tests/test_dvr_gpu.py)
    sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=5, rays_per_frame=30000))
    for pad_mode in (0, 1):
        for thr, order in ((1 << 30, "plain"), (0, "ranked")):
            lib().vidar_dvxlr_set_pad_mode(pad_mode); lib().vidar_dvr_set_sort_min_waves(thr)
            ms = timeit(lambda: dvxlr.render(sigma, origin, points, tindex))
            ms2 = timeit(lambda: dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma))
            ms3 = timeit(lambda: dvr.render_forward(sigma, origin, points, tindex, [5, 16, 200, 200], "train"))
            report(f"A/B pad_mode={pad_mode} order={order} M=150000", ms, render_v2_ms=round(ms2, 4),
                   render_forward_ms=round(ms3, 4))
    sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=1, rays_per_frame=30000))
    for pad_mode in (0, 1):
        lib().vidar_dvxlr_set_pad_mode(pad_mode); lib().vidar_dvr_set_sort_min_waves(1024)
        ms = timeit(lambda: dvxlr.render(sigma, origin, points, tindex), it=30)
        ms2 = timeit(lambda: dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma), it=30)
        report(f"A/B pad_mode={pad_mode} M=30000", ms, render_v2_ms=round(ms2, 4))
    lib().vidar_dvxlr_set_pad_mode(PAD_MODE_DEFAULT); lib().vidar_dvr_set_sort_min_waves(1024)
    for T, rpf in ((1, 30000), (5, 30000)):
        sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=T, rays_per_frame=rpf))
        N, M = tindex.shape
        vol = sigma.numel() * 4
        out = dvxlr.render(sigma, origin, points, tindex)
        cnt = int(((out[3] != 0).any(-1)).sum())
        ms = timeit(lambda: dvxlr.render(sigma, origin, points, tindex))
        report(f"dvxlr.render T={T} M={M}", ms, vol + N * M * 16 + N * M * 4 * (2 + 1026 * 4), samples=cnt)
        em = out[2] * 0.5
        ms = timeit(lambda: dvxlr.get_grad_sigma(em, out[3], tindex, sigma))
        report(f"dvxlr.get_grad_sigma T={T} M={M}", ms, N * M * 1026 * 16 + 2 * vol)
        ms = timeit(lambda: dvr.render_forward(sigma, origin, points, tindex, [T, 16, 200, 200], "train"))
        report(f"dvr.render_forward T={T} M={M}", ms, vol + N * M * 24 + cnt * 4)
        ms = timeit(lambda: dvr.render(sigma, origin, points, tindex, "l1"))
        report(f"dvr.render T={T} M={M}", ms, 2 * vol + N * M * 24 + cnt * 12)
        ms = timeit(lambda: dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma))
        report(f"dvxlr_v2.render_v2 T={T} M={M}", ms, 2 * vol + N * M * 16 + N * M * 4 * (2 + 1026 * 6))
        del out, em


def bench_dvr_trav(which):
    """lane-per-ray kernels (traversal 0) vs the step-parallel kernels (traversal 1, csrc/dvr_par.h) per op and ray
    count; results are identical (tests/test_dvr_gpu.py)."""
    from vidar_amd.synthetic import ray_set
    from vidar_amd.third_lib import dvr, dvxlr, dvxlr_v2
    from vidar_amd._lib import lib
    t = lambda a: torch.from_numpy(a).cuda()
    for T, rpf in ((1, 30000), (2, 30000), (3, 30000), (5, 30000)):
        sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=T, rays_per_frame=rpf))
        N, M = tindex.shape
        vol = sigma.numel() * 4
        lib().vidar_dvr_set_traversal(0)
        cnt = int(((dvxlr.render(sigma, origin, points, tindex)[3] != 0).any(-1)).sum())
        ops = {
            "dvr.render_forward": (lambda: dvr.render_forward(sigma, origin, points, tindex, [T, 16, 200, 200], "train"),
                                   vol + N * M * 24 + cnt * 4),
            "dvr.render": (lambda: dvr.render(sigma, origin, points, tindex, "l1"), 2 * vol + N * M * 24 + cnt * 12),
            "dvxlr.render": (lambda: dvxlr.render(sigma, origin, points, tindex),
                             vol + N * M * 16 + N * M * 4 * (2 + 1026 * 4)),
            "dvxlr_v2.render_v2": (lambda: dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma),
                                   2 * vol + N * M * 16 + N * M * 4 * (2 + 1026 * 6)),
        }
        for name, (fn, nbytes) in ops.items():
            ms = {}
            for trav in (0, 1):
                lib().vidar_dvr_set_traversal(trav)
                ms[trav] = timeit(fn, it=30)
            report(f"{name} M={M}", ms[1], nbytes, lane_per_ray_ms=round(ms[0], 4), step_parallel_ms=round(ms[1], 4),
                   samples=cnt)
    lib().vidar_dvr_set_traversal(-1)


def bench_knn(which):
    from vidar_amd.third_lib.chamferdist import knn_points
    rng = np.random.default_rng(0)
    for P1, P2 in ((30000, 30000), (10000, 30000), (30000, 10000)):
        a = torch.from_numpy(rng.uniform(-50, 50, (1, P1, 3)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.uniform(-50, 50, (1, P2, 3)).astype(np.float32)).cuda()
        ms = timeit(lambda: knn_points(a, b))
        report(f"knn1_d3 {P1}x{P2}", ms, 12 * (P1 + P2) + 12 * P1, Gpairs_s=round(P1 * P2 / ms / 1e6, 1),
               fp32_TFLOPs=round(P1 * P2 * 8 / ms / 1e9, 2))


def bench_msda_coherent(which):
    """both scatter strategies of msda_bwd on spatially coherent queries (what the model produces:
    neighbouring BEV queries sample next to each other; the random reference points of `bench_msda`
    share no lines) + their agreement."""
    from vidar_amd.synthetic import msda_operands_coherent
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import _msda_backward, _msda_forward
    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    for name, B, shapes, Nq, P, px in (("TSA-like", 2, [(200, 200)], 40000, 4, 1.0),
                                       ("SCA-like far (2 level-0 px between queries)", 6, fpn, 10000, 8, 2.0),
                                       ("SCA-like near (8 px)", 6, fpn, 10000, 8, 8.0)):
        value, sh, lsi, loc, w = msda_operands_coherent(0, B, shapes, Nq, P=P, px=px, device="cuda")
        report(f"msda_fwd {name}", timeit(lambda: _msda_forward(value, sh, lsi, loc, w)))
        go = torch.randn(B, Nq, 256, device="cuda")
        outs = {}
        for binned in (False, True):
            outs[binned] = _msda_backward(value, sh, lsi, loc, w, go, binned=binned)
            ms = timeit(lambda: _msda_backward(value, sh, lsi, loc, w, go, binned=binned))
            report(f"msda_bwd {name} binned={binned}", ms)
        for a, b, nm in zip(outs[False], outs[True], ("grad_value", "grad_loc", "grad_w")):
            print(json.dumps({"agreement": nm, "max_abs_diff": float((a - b).abs().max()),
                              "scale": float(a.abs().max())}), flush=True)


def bench_msda(which):
    from vidar_amd.synthetic import msda_operands
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import _msda_forward, _msda_backward
    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    for name, B, shapes, Nq, P in (("TSA", 2, [(200, 200)], 40000, 4), ("SCA", 6, fpn, 10000, 8),
                                   ("Pred", 1, [(200, 200)], 40000, 4)):
        value, sh, lsi, loc, w = msda_operands(0, B, shapes, Nq, P=P, device="cuda")
        L = len(shapes); Nv = value.shape[1]
        fwd_bytes = 4 * (B * Nv * 256 + B * Nq * 8 * L * P * 3 + B * Nq * 256)
        ms = timeit(lambda: _msda_forward(value, sh, lsi, loc, w))
        report(f"msda_fwd {name}", ms, fwd_bytes)
        go = torch.randn(B, Nq, 256, device="cuda")
        for binned in (False, True):
            ms = timeit(lambda: _msda_backward(value, sh, lsi, loc, w, go, binned=binned))
            report(f"msda_bwd {name} binned={binned}", ms, fwd_bytes + 4 * (B * Nq * 256 + B * Nv * 256 + B * Nq * 8 * L * P * 3))


def bench_msda_sca(which):
    """SpatialCrossAttention shape only, few iterations: the driver of the PMC passes (tools/pmc_pass.sh)"""
    from vidar_amd.synthetic import msda_operands
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import _msda_forward, _msda_backward
    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    value, sh, lsi, loc, w = msda_operands(0, 6, fpn, 10000, P=8, device="cuda")
    go = torch.randn(6, 10000, 256, device="cuda")
    report("msda_fwd SCA", timeit(lambda: _msda_forward(value, sh, lsi, loc, w), warm=1, it=3))
    report("msda_bwd SCA binned=True", timeit(lambda: _msda_backward(value, sh, lsi, loc, w, go, binned=True), warm=1, it=3))


def bench_msda_sca_coherent(which):
    """SpatialCrossAttention shape on spatially COHERENT queries (neighbouring queries sample 8 level-0 pixels apart: what
    the model's projected BEV pillars look like), few iterations: the driver of the PMC passes whose traffic is paired
    with the in-model kernel time in bench.py's `roofline.traffic`"""
    from vidar_amd.synthetic import msda_operands_coherent
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import _msda_backward, _msda_forward
    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    import os
    nq = int(os.environ.get("VIDAR_KBENCH_NQ", "10000"))      # 7680 = the padded visible-query count of the bench's step
    value, sh, lsi, loc, w = msda_operands_coherent(0, 6, fpn, nq, P=8, px=8.0, device="cuda")
    go = torch.randn(6, nq, 256, device="cuda")
    report("msda_fwd SCA coherent", timeit(lambda: _msda_forward(value, sh, lsi, loc, w), warm=1, it=3))
    report("msda_bwd SCA coherent binned=True", timeit(lambda: _msda_backward(value, sh, lsi, loc, w, go, binned=True), warm=1, it=3))


def bench_dcn(which):
    """DCNv2 sampling kernels at the backbone shapes (stage 3: 256 ch 58x100 with the 6 gradient images / the 24 history
    images, stage 4: 512 ch 29x50), offsets at init (0) and trained-like (N(0, (1.5 px)^2), vidar_amd/weights.py);
    $VIDAR_KBENCH_DCN_STD overrides the list."""
    import os
    from vidar_amd.plugin.backbones import dcn_col2im
    from vidar_amd._lib import lib, check, ptr, stream_of
    g = torch.Generator().manual_seed(0)
    stds = [float(v) for v in os.environ.get("VIDAR_KBENCH_DCN_STD", "0,1.5,-1.5").split(",")]     # negative: SMOOTH offsets
    for N, C, H, W in ((6, 256, 58, 100), (24, 256, 58, 100), (6, 512, 29, 50)):
        x = torch.randn(N, C, H, W, generator=g).cuda()
        mask = torch.rand(N, 9, H, W, generator=g).cuda()
        gcols = torch.randn(N, C * 9, H * W, generator=g).cuda()
        cols = torch.empty_like(gcols)
        for std in stds:
            off = torch.randn(N, 18, H, W, generator=g)
            if std < 0:       # spatially smooth like a convolution's output (the model's conv_offset): 7x7 box-filtered noise
                off = torch.nn.functional.avg_pool2d(off, 7, stride=1, padding=3)
                off = off / off.std()
            off = (off * abs(std)).cuda()
            tag = f"N={N} C={C} {H}x{W} offsets~{abs(std)}px" + (" smooth" if std < 0 else "")
            ms = timeit(lambda: check(lib().vidar_dcn_im2col_f32(ptr(x), ptr(off), ptr(mask), ptr(cols), N, C, H, W, H, W,
                                                                 3, 3, 1, 1, 1, stream_of(x)), "im2col"))
            report(f"dcn_im2col {tag}", ms, 4 * (x.numel() + cols.numel() + off.numel() + mask.numel()))
            if N > 6:
                continue                      # the history images carry no gradients
            for gather in (False, True):
                ms = timeit(lambda: dcn_col2im(gcols, x, off, mask, 3, 3, 1, 1, 1, H, W, gather=gather))
                report(f"dcn_col2im {tag} gather={gather}", ms,
                       4 * (3 * x.numel() + gcols.numel() + 2 * off.numel() + 2 * mask.numel()))


def bench_affine(which):
    """Frozen BN + (residual) + ReLU of the backbone (csrc/affine_act.hip) at the stage-3 shapes of the 24 history images."""
    from vidar_amd.plugin.backbones import FrozenBN
    for C, res in ((1024, True), (256, False)):
        bn = FrozenBN(C).cuda()
        x = torch.randn(24, C, 58, 100, device="cuda")
        r = torch.randn_like(x) if res else None
        with torch.no_grad():
            ms = timeit(lambda: bn(x, residual=r, relu=True))
        report(f"affine_act_fwd [24,{C},58,100] residual={res}", ms, 4 * x.numel() * (3 if res else 2))


def bench_stem(which):
    """The ResNet stem at the 24 history images: conv1 (MIOpen's 7x7 / stride 2), and BN + ReLU + max-pool as affine_act +
    torch's pooling vs the one-pass kernel the backbone uses."""
    import torch.nn as nn
    import torch.nn.functional as F
    from vidar_amd.plugin.backbones import FrozenBN, stem_bn_relu_pool
    conv = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda()
    conv.weight.requires_grad = False
    bn = FrozenBN(64).cuda()
    x = torch.randn(24, 3, 928, 1600, device="cuda")
    flops = 2 * 24 * 64 * 464 * 800 * 147
    with torch.no_grad():
        ms = timeit(lambda: conv(x))
        print(json.dumps({"op": "stem conv1 7x7 stride 2 (MIOpen)", "ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1)}))
        y = conv(x)
        nb = 4 * (y.numel() + y.numel() // 4)
        report("stem bn+relu+pool two kernels", timeit(lambda: F.max_pool2d(bn(y, relu=True), 3, stride=2, padding=1)), nb)
        report("stem bn+relu+pool one pass", timeit(lambda: stem_bn_relu_pool(y, bn)), nb)


def bench_norm(which):
    """LayerNorm(dropout(x) + residual) on a [40000, 256] BEV map: fused HIP kernels vs torch's three ops."""
    import torch.nn as nn
    import torch.nn.functional as F
    from vidar_amd.plugin.bricks import drop_add_layernorm
    norm = nn.LayerNorm(256).cuda()
    x = torch.randn(1, 40000, 256, device="cuda", requires_grad=True)
    r = torch.randn(1, 40000, 256, device="cuda", requires_grad=True)
    gy = torch.randn(1, 40000, 256, device="cuda")
    nb = x.numel() * 4
    for name, fn in (("fused", lambda: drop_add_layernorm(x, r, norm, 0.1, True)),
                     ("torch", lambda: norm(F.dropout(x, 0.1, True) + r))):
        ms = timeit(lambda: fn())
        report(f"drop_add_ln fwd {name}", ms, 3 * nb)
        y = fn()
        ms = timeit(lambda: torch.autograd.grad(y, [x, r, norm.weight, norm.bias], gy, retain_graph=True))
        report(f"drop_add_ln bwd {name}", ms, 4 * nb)


def bench_lr(which):
    from vidar_amd.plugin.modules.ray_operations.latent_rendering import _PathProb, _RayGather
    for step in (1.0, 0.5):
        occ = torch.randn(1, 200, 200, 16, device="cuda", requires_grad=True)
        a = torch.randn(1, 200, 200, 16, device="cuda", requires_grad=True)
        Q = 40000
        ms = timeit(lambda: _PathProb.apply(occ, 256, step, 0))
        report(f"lr_prob_fwd step={step}", ms, 4 * Q * 32)
        p = _PathProb.apply(occ, 256, step, 0)
        g = torch.randn_like(p)
        ms = timeit(lambda: torch.autograd.grad(p, occ, g, retain_graph=True))
        report(f"lr_prob_bwd step={step}", ms, 4 * Q * 48)
        pd = p.detach().requires_grad_(True)
        ms = timeit(lambda: _RayGather.apply(pd, a, 256, step, 1e-3))
        report(f"lr_gather_fwd step={step}", ms, 4 * Q * 64)
        f = _RayGather.apply(pd, a, 256, step, 1e-3)
        ms = timeit(lambda: torch.autograd.grad(f, [pd, a], g, retain_graph=True))
        report(f"lr_gather_bwd step={step}", ms, 4 * Q * 112)


def bench_ray(which):
    from vidar_amd.plugin.dense_heads.ray_ops import ray_ce, ray_gumbel, gumbel_noise
    from vidar_amd.synthetic import ray_set
    sig, origin, points, tindex = ray_set(seed=0, N=1, T=1, rays_per_frame=30000)
    sigma = torch.randn(1, 16, 200, 200, device="cuda", requires_grad=True)
    o, p, ti = (torch.from_numpy(a[0]).cuda() for a in (origin, points, tindex))
    ms = timeit(lambda: ray_ce(sigma, o, p, ti))
    report("ray_ce_fwd P=30000", ms, 4 * (16 * 200 * 200 + 30000 * 5), Mrays_s=round(30000 / ms / 1e3, 1))
    ce, valid = ray_ce(sigma, o, p, ti)
    g = torch.ones_like(ce)
    ms = timeit(lambda: torch.autograd.grad(ce, sigma, g, retain_graph=True))
    report("ray_ce_bwd P=30000", ms, 4 * (2 * 16 * 200 * 200 + 30000 * 5))
    from vidar_amd.synthetic import dense_rays
    pts, tix = dense_rays(1, 16, 200, 200, "cuda")
    noise = gumbel_noise(pts.shape[0], 512)
    ms = timeit(lambda: ray_gumbel(sigma, o, pts, tix, noise))
    report(f"ray_gumbel_fwd R={pts.shape[0]}", ms, 4 * (16 * 200 * 200 + pts.shape[0] * 516))
    d = ray_gumbel(sigma, o, pts, tix, noise)
    ms = timeit(lambda: torch.autograd.grad(d, sigma, torch.ones_like(d), retain_graph=True))
    report(f"ray_gumbel_bwd R={pts.shape[0]}", ms, 4 * (2 * 16 * 200 * 200 + pts.shape[0] * 4))


def bench_gemm(which):
    """the hot path's GEMM shapes: library (torch addmm / bmm, TunableOp off) vs csrc/gemm_mfma.hip in both modes.
    TF/s is the effective fp32-product rate 2MNK / time."""
    from vidar_amd import gemm as G
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1

    def line(name, flops, nbytes, **ms):
        d = {"op": name}
        for k, v in ms.items():
            d[k + "_ms"] = round(v, 4); d[k + "_TF"] = round(flops / v / 1e9, 1)
        d["min_HBM_ms"] = round(nbytes / 8e9, 4)
        print(json.dumps(d), flush=True)

    # nn.Linear shapes: value_proj (SCA), TSA offsets, FFN, output_proj
    for (M, K, N, tag) in ((184950, 256, 256, "value_proj SCA"), (80000, 256, 256, "value_proj TSA"),
                           (40000, 512, 128, "tsa sampling_offsets"), (40000, 256, 512, "ffn.0"),
                           (40000, 512, 256, "ffn.1"), (46080, 256, 512, "sca sampling_offsets")):
        x, w, b = rnd(M, K), rnd(N, K) * 0.1, rnd(N)
        gy = rnd(M, N)
        fl = 2.0 * M * N * K
        ms = dict(lib=timeit(lambda: torch.addmm(b, x, w.t())))
        for m, p in (("f32", G.F32), ("bf16x3", G.BF16X3)):
            ms[m] = timeit(lambda: G.linear_forward(x, w, b, False, p))
        line(f"linear fwd [{M},{K}]x[{K},{N}] {tag}", fl, 4 * (M * K + M * N + N * K), **ms)
        ms = dict(lib=timeit(lambda: gy @ w))
        for m, p in (("f32", G.F32), ("bf16x3", G.BF16X3)):
            ms[m] = timeit(lambda: G.linear_grad_input(gy, w, p))
        line(f"linear dx  [{M},{N}]x[{N},{K}] {tag}", fl, 4 * (M * K + M * N + N * K), **ms)
        ms = dict(lib=timeit(lambda: (x.t() @ gy).t()))
        for m, p in (("f32", G.F32), ("bf16x3", G.BF16X3)):
            ms[m] = timeit(lambda: G.linear_grad_weight(gy, x, p))
        line(f"linear dw  [{N},{M}]x[{M},{K}] {tag}", fl, 4 * (M * K + M * N + N * K), **ms)
    # backbone: out[b] = W [Co,Ci] x[b] [Ci,HW]
    for (Bn, Co, Ci, HW, tag) in ((24, 1024, 256, 5800, "layer3 conv3"), (24, 256, 1024, 5800, "layer3 conv1"),
                                  (24, 256, 2304, 5800, "layer3 dcn"), (24, 512, 128, 23200, "layer2 conv3"),
                                  (24, 2048, 512, 1450, "layer4 conv3")):
        w, x = rnd(Co, Ci) * 0.05, rnd(Bn, Ci, HW)
        sc, sh, res = rnd(Co), rnd(Co), rnd(Bn, Co, HW)
        fl = 2.0 * Bn * Co * Ci * HW
        nb = 4 * (Bn * Ci * HW + Bn * Co * HW)
        ms = dict(lib=timeit(lambda: torch.bmm(w.view(1, Co, Ci).expand(Bn, -1, -1), x)))
        for m, p in (("f32", G.F32), ("bf16x3", G.BF16X3)):
            ms[m] = timeit(lambda: G.conv_forward(w, x, precision=p))
            ms[m + "_bn_res_relu"] = timeit(lambda: G.conv_forward(w, x, sc, sh, res, True, p))
        line(f"conv fwd {Bn}x[{Co},{Ci}]x[{Ci},{HW}] {tag}", fl, nb, **ms)
        gy = rnd(6, Co, HW); x6 = x[:6]
        fl6 = fl / 4
        ms = dict(lib=timeit(lambda: torch.bmm(w.view(1, Co, Ci).transpose(1, 2).expand(6, -1, -1), gy)))
        for m, p in (("f32", G.F32), ("bf16x3", G.BF16X3)):
            ms[m] = timeit(lambda: G.conv_grad_input(w, gy, p))
        line(f"conv dx  6x[{Ci},{Co}]x[{Co},{HW}] {tag}", fl6, nb / 4, **ms)
        ms = dict(lib=timeit(lambda: torch.bmm(gy, x6.transpose(1, 2)).sum(0)))
        for m, p in (("f32", G.F32), ("bf16x3", G.BF16X3)):
            ms[m] = timeit(lambda: G.conv_grad_weight(gy, x6, p))
        line(f"conv dw  6x[{Co},{HW}]x[{HW},{Ci}] {tag}", fl6, nb / 4, **ms)


def bench_conv_offset(which):
    """the DCNv2 `conv_offset` (3x3, C -> 27) at the backbone's four cases: csrc/conv3x3_mfma.hip vs the library convolution"""
    import torch.nn.functional as F
    from vidar_amd.plugin.backbones import _Conv3x3Few
    torch.manual_seed(0)
    for (N, C, H, W) in [(24, 256, 58, 100), (6, 256, 58, 100), (24, 512, 29, 50), (6, 512, 29, 50)]:
        x = torch.randn(N, C, H, W, device="cuda")
        w = torch.randn(27, C, 3, 3, device="cuda") * 0.02
        b = torch.randn(27, device="cuda")
        flops = 2 * 27 * C * 9 * N * H * W
        with torch.no_grad():
            ref = F.conv2d(x, w, b, padding=1)
            got = _Conv3x3Few.apply(x, w, b)
            err = float((got - ref).abs().max() / ref.abs().max())
            ms_lib = timeit(lambda: F.conv2d(x, w, b, padding=1))
            ms_own = timeit(lambda: _Conv3x3Few.apply(x, w, b))
        report(f"conv_offset[{N},{C},{H},{W}] own", ms_own, 4 * (x.numel() + got.numel()), TFLOPs=round(flops / ms_own / 1e9, 1),
               frac_fp32_mfma=round(flops * 32 / 27 / ms_own / 1e9 / 157.3, 3), rel_err_vs_lib=err)
        report(f"conv_offset[{N},{C},{H},{W}] library", ms_lib, 4 * (x.numel() + got.numel()), TFLOPs=round(flops / ms_lib / 1e9, 1))


def bench_gemm_pmc(which):
    """a few launches of the representative GEMM shapes per mode, for the counter passes (tools/pmc_pass.sh)"""
    from vidar_amd import gemm as G
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
    x, w, b = rnd(184950, 256), rnd(256, 256) * 0.1, rnd(256)
    wc, xc = rnd(1024, 256) * 0.05, rnd(24, 256, 5800)
    for p in (G.F32, G.BF16X3):
        for _ in range(3):
            G.linear_forward(x, w, b, False, p)
            G.conv_forward(wc, xc, precision=p)
    torch.cuda.synchronize()


if __name__ == "__main__":
    import os
    if os.environ.get("VIDAR_MSDA_ITEM_ORDER") is not None:          # A/B of the gather kernels' item order (0 banded, 1 head-major)
        from vidar_amd._lib import lib
        lib().vidar_msda_set_item_order(int(os.environ["VIDAR_MSDA_ITEM_ORDER"]))
    if os.environ.get("VIDAR_GEMM_VARIANT") is not None:               # A/B of the GEMM kernel's structure (0 classic, 1 / 2 wave-specialised)
        from vidar_amd._lib import lib
        lib().vidar_gemm_set_variant(int(os.environ["VIDAR_GEMM_VARIANT"]))
    which = sys.argv[1:] or ["dvr", "knn", "msda", "lr", "ray"]
    print(json.dumps({"device": torch.cuda.get_device_name(0)}))
    for w in which:
        globals()["bench_" + w](w)
