"""bench.py -- train samples/s of ViDAR's hot path (6-cam images -> ResNet101-DCNv2 + FPN -> BEV encode ->
latent render -> occupancy head -> ray-march / chamfer losses -> backward -> AdamW) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU, DDP gradient all-reduce over RCCL.  Prints ONE JSON line on rank 0 (the last line of
its stdout) with the driver contract fields plus
  `roofline`         the dominant HIP kernel of the SURVEY 8(a) path, timed live with HIP events on the launch stream
                     inside the timed region; `traffic` = calibrated PMC bytes (profiles/pmc_traffic.json);
  `configs`          short records of BASELINE.json's other named configs, timed in the same process (every N);
  `ddp`              DDP's bucket accounting incl. the buckets it rebuilds after the first step (N > 1);
  `roofline_kernels` per-kernel rooflines at the BASELINE shapes (dvr family up to the c4 stress shape, MSDA, KNN; N = 1);
  `cpu_baseline`     op-level, full-size CPU rows of the oracle port next to the GPU's own times (rank 0, N = 1, a
                     child process outside the timed region).
A "sample" = one 5-frame x 6-camera sequence (global batch = number of GPUs x --samples-per-gpu; the reference asserts
1 sample per GPU: detectors/vidar.py:306)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KB -> bytes; FETCH not
# doubled: the x2 rule of the microarch guide is calibrated for wide coalesced streams only).  These
# are NOT measured by this process: they are read back from the committed PMC summaries of a kbench run
# at the same shapes and labelled with their source file in the JSON line (`traffic_source`).
PMC_TRAFFIC = ROOT / "profiles" / "pmc_traffic.json"     # {op name: {"bytes": ..., "source": "profiles/..."}}


def pmc_traffic(name):
    try:
        rec = json.loads(PMC_TRAFFIC.read_text()).get(name)
    except (OSError, ValueError):
        rec = None
    return (rec["bytes"], rec["source"]) if rec else (None, None)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="vidar_1_8_nusc_1future")
    ap.add_argument("--rays-per-frame", type=int, default=30000)
    ap.add_argument("--samples-per-gpu", type=int, default=1,
                    help="per-GPU batch (the reference is fixed at 1, vidar.py:306; BASELINE config 3 sizes it to HBM)")
    ap.add_argument("--no-backbone", action="store_true",
                    help="feed FPN pyramids instead of images (hot path of SURVEY 8a only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ddp-ab", action="store_true",
                    help="N > 1: skip the short records of the other gradient-exchange modes (flat / flat2 / torch, `ddp_ab`)")
    ap.add_argument("--cpu-baseline-step", action="store_true",
                    help="also time the oracle port of the WHOLE step on a bounded sample (BEV 50x50; round-2 leg)")
    ap.add_argument("--weights", default="trained_like",
                    help="weight state of the timed model: `trained_like` (default: DCNv2 offsets ~N(0,(1.5 px)^2) with "
                         "non-uniform masks, per-query deformable-attention offsets / logits -- the access pattern of a "
                         "model that starts from the pretrained backbone the reference always loads, config :400; "
                         "vidar_amd/weights.py), `init` (what init_weights leaves: zero DCNv2 offsets, one sampling ring "
                         "per head -- the gather / scatter kernels' best case, rounds 1-5), or a `.pth` path (mmcv layout)")
    ap.add_argument("--extra-configs",
                    default="vidar_1_8_nusc_1future+init!,vidar_1_8_nusc_3future!,vidar_1_8_nusc_1future@1:bf16x3!,"
                            "vidar_1_8_nusc_3future+init,"
                            "mem_efficient_vidar_1_8_nusc_3future,vidar_OpenScene_mini_full_3future,"
                            "vidar_1_8_nusc_3future@1:bf16x3,"
                            "vidar_full_nusc_1future@2,vidar_full_nusc_1future@4,vidar_full_nusc_1future@8",
                    help="comma-separated `config[@samples_per_gpu][:gemm][+weights][!]` entries timed after the main one in the same "
                         "run (short records under `configs`, each with its peak device memory): BASELINE.json's other "
                         "named configs -- the north star's target sentence names vidar_1_8_nusc_3future (and the "
                         "reference's memory-efficient variant of it, README.md:143-148); OpenScene = 8 cameras; "
                         ":bf16x3 = the split-bf16 MFMA GEMM path (a labelled second record, the headline stays fp32); "
                         "@2/@4/@8 = per-GPU batch sweep of config c3 (\"per-GPU batch sized to 288 GB\": 8 samples "
                         "are ~260 GB); a trailing ! = timed with the headline's --steps / --warmup instead of "
                         "--extra-steps / --extra-warmup (the north star's target config and the bf16x3 record)")
    ap.add_argument("--gemm", choices=["lib", "auto", "f32", "bf16x3"], default=None,
                    help="what the Linear / 1x1-convolution products of the main record run on (default: vidar_amd.gemm."
                         "mode(), i.e. $VIDAR_GEMM or the package default); extra configs take it as `name@spg:gemm`")
    ap.add_argument("--extra-budget-s", type=float, default=270.0,
                    help="wall-clock budget of the extra configs together: entries that would start after it are recorded "
                         "as skipped (the contract line must appear within minutes)")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--extra-warmup", type=int, default=2)
    ap.add_argument("--cpu-baseline-only", choices=["ops", "step", "full"], help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=min(os.cpu_count() or 1, 16))
    ap.add_argument("--cpu-baseline-timeout", type=int, default=600)
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time ONE full-size CPU step instead of the bounded sample (needs > 150 GB of host RAM "
                         "and ~10 min; it took down a 1-GPU pool box, so it is opt-in)")
    ap.add_argument("--no-kernel-rooflines", action="store_true",
                    help="skip the per-kernel roofline list (dvr family + MSDA at BASELINE shapes)")
    ap.add_argument("--op-table", action="store_true", help="print the per-op timing table to stderr")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="leave the library GEMMs on their default heuristics (A/B of vidar_amd/gemm_tuning.py)")
    ap.add_argument("--tunableop-file", help="where TunableOp writes the solutions it found (default /tmp/...)")
    return ap.parse_args()


def relaunch_command(args, argv, port=None):
    """`python bench.py --gpus N` without a launcher: the command that re-runs this script as N ranks of ONE node
    (what the driver's own multi-GPU command looks like), or None when no re-launch is needed.  A rank environment
    (WORLD_SIZE from torch.distributed.run) always wins -- then --gpus must agree with it."""
    if "WORLD_SIZE" in os.environ or args.cpu_baseline_only:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if not args.cpu_baseline_only and args.gpus != world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        return None
    if args.gpus <= 1:
        return None
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), *argv]


def synthetic_images(seed, T, num_cams, hw, device, scale=1):
    """N(0,1) images, seeded per (rank, sample).  Drawn ON the device they are used on: host-side generation
    of a per-GPU batch of 8 (4.3 GB) took longer than timing it."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(1, T, num_cams, 3, hw[0] // scale, hw[1] // scale, generator=g, device=dev)


def cpu_baseline(config, threads, with_backbone=True, reduced=True):
    """The oracle port of the same training step on host cores.  Default: a BOUNDED sample -- BEV 50x50 (1/16 of
    the queries), quarter-resolution images (1/16 of the pixels), 1 875 rays/frame -- one warm-up + one timed
    step (~10 s), scaled x16 to the metric's unit and labelled as such (the chamfer term grows x256, so the true
    full-size CPU rate is lower than reported).  reduced=False times ONE full-size step instead: it needs well
    over 100 GB of host RAM (fp32 ResNet101 activations of 12 full-resolution images + oracle intermediates)."""
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import fpn_features, make_sample
    torch.set_num_threads(threads)
    div = 4 if reduced else 1
    cfg = get_config(config, bev_h=200 // div, bev_w=200 // div, with_backbone=with_backbone)
    torch.manual_seed(0); np.random.seed(0)
    model = T.build_model(cfg).train()
    opt = T.build_optimizer(model)
    metas, gt = make_sample(0, rays_per_frame=30000 // (div * div), future_frames=cfg["future_frames"],
                            num_cams=cfg["num_cams"], img_hw=cfg["img_hw"])
    if with_backbone:
        qhw = (cfg["img_hw"][0] // div, cfg["img_hw"][1] // div)
        for m in metas:
            m["img_shape"] = [(qhw[0], qhw[1], 3)] * cfg["num_cams"]
            k = np.diag([1.0 / div, 1.0 / div, 1.0, 1.0])
            m["lidar2img"] = [k @ a for a in m["lidar2img"]]
        batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt)],
                     img=synthetic_images(0, 5, cfg["num_cams"], cfg["img_hw"], "cpu", scale=div))
    else:
        shapes = [((h + div - 1) // div, (w + div - 1) // div) for h, w in cfg["fpn_shapes"]]
        feats = fpn_features(0, 5, num_cams=cfg["num_cams"], shapes=shapes)
        batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt)], img_feats=feats)
    with cpu_ops.patched():
        if reduced:
            T.train_step(model, opt, batch)         # warm-up (cheap at this size)
        t0 = time.perf_counter()
        T.train_step(model, opt, batch)
        dt = time.perf_counter() - t0
    what = "with" if with_backbone else "without"
    if reduced:
        return dict(value=1.0 / (dt * 16.0), unit="samples/s", cores=threads, kind="port",
                    sample=f"bounded sample: oracle port of the step ({what} backbone) at BEV 50x50 (1/16 of the "
                           f"queries), 1/4-res images (1/16 of the pixels), 1875 rays/frame: {dt:.2f} s/step measured "
                           f"on {threads} threads, value = 1/(16 x that); the O(N*M) chamfer term grows x256, so the "
                           f"full-size CPU rate is lower (a full-size step needs > 100 GB of host RAM: --cpu-baseline-full)")
    return dict(value=1.0 / dt, unit="samples/s", cores=threads, kind="port",
                sample=f"oracle port of ONE full-size training step ({what} backbone; BEV 200x200, "
                       f"{cfg['num_cams']}x{cfg['img_hw'][0]}x{cfg['img_hw'][1]} images, 30000 rays/frame, first step, "
                       f"no warm-up): {dt:.1f} s measured on {threads} torch threads, peak RSS {_peak_rss_gb():.0f} GB")


def usable_cores():
    """cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a GPU box reports 256
    logical CPUs while the job's cgroup grants far fewer -- 256 OpenMP threads then run 8 x SLOWER than 16)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def _peak_rss_gb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def _once_or_twice(fn):
    """seconds of fn(): the second of two calls, unless the first already took > 1 s (then that one)"""
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    if dt > 1.0:
        return dt
    t0 = time.perf_counter(); fn()
    return time.perf_counter() - t0


def cpu_baseline_ops(threads):
    """the hot-path ops of SURVEY 8(a) at FULL size on host cores, one row per op --
    the oracle's pure-PyTorch restatements (MSDA formula, LatentRendering, GT-ray march + CE, gumbel render, naive
    O(N*M) chamfer) on `threads` torch threads, the reference's own knn_cpu.cpp build (single-threaded by
    construction) and the OpenMP C restatement of the dvr / dvxlr kernels.  Every input fits in host RAM at
    full size; the rows take ~20-40 s together.  -> dict(ops=[...], cores=threads)"""
    from oracle import chamfer as C
    from oracle import cpu_ops
    from oracle import dvr as O
    from oracle import latent_render as LR
    from oracle import msda as M
    from vidar_amd.synthetic import dense_rays, msda_operands, ray_set
    torch.set_num_threads(threads)
    rows = []

    def add(op, seconds, shape, impl, thr=threads):
        rows.append(dict(op=op, cpu_ms=round(seconds * 1e3, 2), shape=shape, impl=impl, threads=thr))

    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    formula = "pure-PyTorch MSDA formula (per-level F.grid_sample), the reference's CPU branch restated (oracle/msda.py)"
    for tag, B, shapes, Nq, P in (("L=1,P=4", 2, [(200, 200)], 40000, 4), ("L=4,P=8", 6, fpn, 10000, 8)):
        value, sh, lsi, loc, w = msda_operands(0, B, shapes, Nq, P=P)
        shape = f"B={B} Nv={value.shape[1]} Nq={Nq} L={len(shapes)} P={P}"
        with torch.no_grad():
            add(f"msda_fwd[{tag}]", _once_or_twice(lambda: M.msda_grid_sample(value, sh, loc, w)), shape, formula)
        v, l, ww = value.requires_grad_(), loc.requires_grad_(), w.requires_grad_()
        out = M.msda_grid_sample(v, sh, l, ww)
        go = torch.randn_like(out)
        add(f"msda_bwd[{tag}]", _once_or_twice(lambda: torch.autograd.grad(out, [v, l, ww], go, retain_graph=True)),
            shape, formula + ", autograd backward")
        del value, loc, w, v, l, ww, out, go
    lr_impl = "latent_rendering.py:79-162 restated with torch ops on CPU tensors (oracle/latent_render.py)"
    occ = torch.randn(1, 200, 200, 16, requires_grad=True)
    a = torch.randn(1, 200, 200, 16, requires_grad=True)
    with torch.no_grad():
        add("lr_prob_fwd", _once_or_twice(lambda: LR.path_prob(occ, 256, 1.0, "sigmoid")), "bev 200x200 x 16 bins, 257 waypoints", lr_impl)
    p = LR.path_prob(occ, 256, 1.0, "sigmoid")
    g = torch.randn_like(p)
    add("lr_prob_bwd", _once_or_twice(lambda: torch.autograd.grad(p, occ, g, retain_graph=True)), "same", lr_impl + ", autograd backward")
    pd = p.detach().requires_grad_(True)
    with torch.no_grad():
        add("lr_gather_fwd", _once_or_twice(lambda: LR.gather(pd, a, 256, 1.0, 1e-3)), "same", lr_impl)
    f = LR.gather(pd, a, 256, 1.0, 1e-3)
    add("lr_gather_bwd", _once_or_twice(lambda: torch.autograd.grad(f, [pd, a], g, retain_graph=True)), "same", lr_impl + ", autograd backward")
    del occ, a, p, pd, f, g
    sig, origin, points, tindex = ray_set(seed=0, N=1, T=1, rays_per_frame=30000)
    sigma = torch.randn(1, 16, 200, 200, requires_grad=True)
    o, pts, ti = (torch.from_numpy(x[0]) for x in (origin, points, tindex))
    head_impl = "vidar_head_base.py:420-509,:586-592 restated (trilinear grid_sample of 513 waypoints per ray + CE; oracle/head.py)"
    with torch.no_grad():
        add("ray_ce_fwd", _once_or_twice(lambda: cpu_ops._ray_ce(sigma, o, pts, ti)), "30000 rays x 513 waypoints, volume 16x200x200", head_impl)
    ce, _ = cpu_ops._ray_ce(sigma, o, pts, ti)
    add("ray_ce_bwd", _once_or_twice(lambda: torch.autograd.grad(ce, sigma, torch.ones_like(ce), retain_graph=True)), "same", head_impl + ", autograd backward")
    dp, dt_ = dense_rays(1, 16, 200, 200)
    noise = cpu_ops._gumbel_noise(dp.shape[0], 512)
    gum_impl = "vidar_head_base.py:594-630,:754-773 restated (hard gumbel-softmax hit + straight-through mass; oracle/head.py)"
    with torch.no_grad():
        add("ray_gumbel_fwd", _once_or_twice(lambda: cpu_ops._ray_gumbel(sigma, o, dp, dt_, noise)), f"{dp.shape[0]} dense rays x 512 waypoints", gum_impl)
    d = cpu_ops._ray_gumbel(sigma, o, dp, dt_, noise)
    add("ray_gumbel_bwd", _once_or_twice(lambda: torch.autograd.grad(d, sigma, torch.ones_like(d), retain_graph=True)), "same", gum_impl + ", autograd backward")
    del ce, d
    rng = np.random.default_rng(0)
    src = torch.from_numpy(rng.uniform(-5, 5, (1, 10000, 3)).astype(np.float32))
    dst = torch.from_numpy(rng.uniform(-5, 5, (1, 30000, 3)).astype(np.float32))
    add("knn1_d3_fwd", _once_or_twice(lambda: cpu_ops._knn_points(src, dst)), "training chamfer: 10000 rendered x 30000 GT points, one direction",
        "naive O(N*M) dense-expand form of mmdet3d chamfer_distance (call site vidar_head_base.py:654)")
    # evaluation chamfer: the reference's own extension when oracle/_ref travelled with the snapshot
    a3 = rng.uniform(-50, 50, (1, 30000, 3)).astype(np.float32)
    b3 = rng.uniform(-50, 50, (1, 30000, 3)).astype(np.float32)
    try:
        from oracle import build_ref
        ref = build_ref.load("ref_chamferdist_C")
        l3 = torch.tensor([30000])
        t0 = time.perf_counter()
        ref.knn_points_idx(torch.from_numpy(a3), torch.from_numpy(b3), l3, l3, 1, -1)
        add("chamferdist.knn_points_idx[30000x30000]", time.perf_counter() - t0, "evaluation CD, one direction",
            "the reference's own ext.cpp + knn_cpu.cpp compiled unmodified (oracle/_ref), single-threaded by construction", thr=1)
    except Exception:
        t0 = time.perf_counter()
        C.knn_points_idx(a3, b3)
        add("chamferdist.knn_points_idx[30000x30000]", time.perf_counter() - t0, "evaluation CD, one direction",
            "numpy restatement of knn_cpu.cpp:7-58 (oracle/chamfer.py; oracle/_ref not present on this box)", thr=1)
    O.set_threads(threads)
    dvr_impl = "plain-C restatement of the kernel bodies, fp64 in the reference's operation order, OpenMP over rays (oracle/dvr_oracle.c)"
    add("dvxlr.render[M=30000]", _once_or_twice(lambda: O.dvxlr_render(sig, origin, points, tindex)), "30000 rays, volume 16x200x200", dvr_impl)
    add("dvr.render_forward[M=30000]", _once_or_twice(lambda: O.render_forward(sig, origin, points, tindex, "train")), "same", dvr_impl)
    add("dvr.render[M=30000]", _once_or_twice(lambda: O.render(sig, origin, points, tindex, "l1")), "same", dvr_impl)
    # second column: the rows that scale with cores (OpenMP over rays) on ALL host cores (BASELINE.md section 3 asks
    # for the box's cores; `threads` = 16 is where the torch rows stop scaling -- the MSDA formula on 256 torch threads
    # measured 5 x SLOWER than on 16, profiles/r04_bench_driver_like.json of the first try, so it is not repeated)
    allc = usable_cores()
    if allc > threads:
        by = {r["op"]: r for r in rows}
        O.set_threads(allc)
        for op, fn in (("dvxlr.render[M=30000]", lambda: O.dvxlr_render(sig, origin, points, tindex)),
                       ("dvr.render_forward[M=30000]", lambda: O.render_forward(sig, origin, points, tindex, "train")),
                       ("dvr.render[M=30000]", lambda: O.render(sig, origin, points, tindex, "l1"))):
            by[op]["cpu_ms_all_cores"] = round(_once_or_twice(fn) * 1e3, 2)
        O.set_threads(threads)
    return dict(ops=rows, cores=threads, all_cores=allc)


def cpu_baseline_subprocess(args, mode="ops"):
    """Run a CPU leg in a child with a hard wall-clock limit so it can never stall the bench.
    mode: "ops" (op-level full-size rows), "step" (bounded whole-step sample), "full" (one full-size step)."""
    import subprocess
    base = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", mode, "--config", args.config,
            "--cpu-threads", str(args.cpu_threads)] + (["--no-backbone"] if args.no_backbone else [])
    env = dict(os.environ, OMP_NUM_THREADS=str(args.cpu_threads), MKL_NUM_THREADS=str(args.cpu_threads))
    limit = args.cpu_baseline_timeout if mode == "full" else 300
    try:
        r = subprocess.run(base, capture_output=True, text=True, timeout=limit, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        note = "cpu leg produced no result: " + r.stderr.strip()[-200:]
    except subprocess.TimeoutExpired:
        note = f"cpu leg ({mode}) exceeded {limit} s and was stopped"
    return dict(value=None, unit="samples/s", cores=args.cpu_threads, kind="port", sample=note)


def cpu_baseline_record(args, gpu_ops, steps, kernel_rows):
    """`cpu_baseline` of the JSON line: op-level full-size CPU rows next to the GPU's own numbers for the same op,
    and their call-count-weighted sum = a LOWER BOUND on the full-size CPU step (only the 8(a) ops are in it: no
    GEMMs, no image backbone, no elementwise work) -> `value` = 1 / that bound, an UPPER bound on CPU samples/s."""
    rec = cpu_baseline_subprocess(args, "ops")
    if "ops" not in rec:
        return rec
    gpu_kernels = {r["kernel"]: r["avg_ms"] for r in (kernel_rows or [])}
    bound_ms = 0.0
    for r in rec["ops"]:
        g = gpu_ops.get(r["op"])
        if g is not None:
            r["calls_per_step"] = g["calls"] / steps
            r["gpu_ms"] = round(g["avg_ms"], 4)
            bound_ms += r["cpu_ms"] * r["calls_per_step"]
        elif r["op"] in gpu_kernels:
            r["calls_per_step"] = 0                      # surface-only / evaluation op: not on the training step
            r["gpu_ms"] = gpu_kernels[r["op"]]
        if r.get("gpu_ms"):
            r["speedup"] = round(r["cpu_ms"] / r["gpu_ms"], 1)
            if r.get("cpu_ms_all_cores"):
                r["speedup_all_cores"] = round(r["cpu_ms_all_cores"] / r["gpu_ms"], 1)
    out = dict(value=(1e3 / bound_ms) if bound_ms > 0 else None, unit="samples/s", cores=rec["cores"], kind="port",
               host_cores=os.cpu_count(), usable_cores=usable_cores(), all_cores_column=rec.get("all_cores"),
               step_lower_bound_ms=round(bound_ms, 1), ops=rec["ops"],
               sample=(f"op-level, FULL size (BEV 200x200, 6 x 30825 px, 30000 rays): each SURVEY 8(a) op's oracle port "
                       f"timed once on {rec['cores']} threads (knn_cpu: 1), weighted by the GPU step's own launches per "
                       f"step; their sum ({bound_ms / 1e3:.1f} s) is a LOWER bound on the CPU step -- GEMMs, the image "
                       f"backbone and elementwise work are not in it -- so `value` is an UPPER bound on CPU samples/s"))
    if args.cpu_baseline_step or args.cpu_baseline_full:
        out["whole_step"] = cpu_baseline_subprocess(args, "full" if args.cpu_baseline_full else "step")
    return out


def hip_time(fn, iters=10, warm=2):
    """average ms of fn() with HIP events on torch's current stream (the stream every op launches on)."""
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_rooflines(dev):
    """Per-kernel roofline list at the BASELINE shapes, timed live in this process (outside the timed
    step region): the dvr / dvxlr ray-march family the north star names (30 000 = one frame, 150 000 = a
    5-frame sample) and the MSDA gather / scatter.  achieved = SURVEY 8d algorithmic bytes / avg ms."""
    from vidar_amd.plugin.modules.multi_scale_deformable_attn_function import (
        _msda_backward, _msda_forward, msda_bwd_bytes, msda_fwd_bytes)
    from vidar_amd.synthetic import ray_set
    from vidar_amd.third_lib import dvr, dvxlr, dvxlr_v2
    rows = []

    def add(kernel, ms, nbytes, bound="hbm", note=None):
        gbps = nbytes / ms / 1e6
        r = dict(kernel=kernel, avg_ms=round(ms, 4), bytes=int(nbytes), achieved=round(gbps, 1), unit="GB/s",
                 peak=HBM_PEAK_GBPS, frac=round(gbps / HBM_PEAK_GBPS, 4), bound=bound)
        if note:
            r["note"] = note
        rows.append(r)

    t = lambda a: torch.from_numpy(a).to(dev)
    # one frame, a 5-frame sample, and SURVEY 8d's OpenScene stress shape (config c4: T = 4 + 6 frames,
    # 9 x 30 000 rays, origins up to 30 voxels off centre)
    for T_, rpf, jitter in ((1, 30000, 0.0), (5, 30000, 0.0), (10, 27000, 30 * 0.512)):
        sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=T_, rays_per_frame=rpf, origin_jitter=jitter))
        N, M = tindex.shape
        vol = sigma.numel() * 4
        out = dvxlr.render(sigma, origin, points, tindex)
        live = (out[3] != 0).any(-1)                          # [N, M, 1026]: the samples of every ray
        cnt = int(live.sum())                                 # traversed voxels (dvr bytes depend on it)
        longest = int(live.sum(-1).max())                     # samples of the longest ray
        add(f"dvxlr.render[M={M}]", hip_time(lambda: dvxlr.render(sigma, origin, points, tindex)),
            vol + N * M * 16 + N * M * 4 * (2 + 1026 * 4), note="16.4 KB/ray of API-mandated padded rows")
        em = out[2] * 0.5
        add(f"dvxlr.get_grad_sigma[M={M}]", hip_time(lambda: dvxlr.get_grad_sigma(em, out[3], tindex, sigma)),
            N * M * 1026 * 16 + 2 * vol)
        add(f"dvxlr_v2.render_v2[M={M}]", hip_time(lambda: dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma)),
            2 * vol + N * M * 16 + N * M * 4 * (2 + 1026 * 6))
        add(f"dvr.render_forward[M={M}]",
            hip_time(lambda: dvr.render_forward(sigma, origin, points, tindex, [T_, 16, 200, 200], "train")),
            vol + N * M * 24 + cnt * 4,
            bound=("step-parallel traversal (per-axis tMax chains + lane-per-step integration), instruction issue, not HBM"
                   if N * M <= 24576 else "fp64 issue (sequential DDA per lane), not HBM"))
        # the bound as a NUMBER: a ray is a serial chain of fp64 traversal steps, a wave lasts as long as its longest ray, and
        # with <= 1 wave per SIMD (469 waves at 30 000 rays) the launch lasts as long as its longest wave
        rows[-1]["serial_chain"] = dict(longest_ray_steps=longest, total_steps=cnt,
                                        us_per_step_of_longest_ray=round(rows[-1]["avg_ms"] * 1e3 / max(longest, 1), 3),
                                        waves=(N * M + 63) // 64, simds=1024)
        # ... and as a MEASUREMENT: the same entry point on the longest ray ALONE is the launch's critical path (nothing
        # can finish before its slowest ray has); launch time / that = how far the 30 000-ray launch is from the
        # latency floor of its own traversal -- the figure this kernel is held to instead of an HBM fraction
        j = int(live.sum(-1).reshape(-1).argmax())
        n_, m_ = divmod(j, M)
        p1, t1 = points[n_:n_ + 1, m_:m_ + 1].contiguous(), tindex[n_:n_ + 1, m_:m_ + 1].contiguous()
        s1, o1 = sigma[n_:n_ + 1].contiguous(), origin[n_:n_ + 1].contiguous()
        one = hip_time(lambda: dvr.render_forward(s1, o1, p1, t1, [T_, 16, 200, 200], "train"), iters=20)
        rows[-1]["latency_model"] = dict(
            longest_ray_alone_ms=round(one, 4), launch_over_longest_ray=round(rows[-1]["avg_ms"] / one, 2),
            note="a launch cannot beat its longest ray: `longest_ray_alone_ms` (one ray, same entry point, includes the "
                 "launch overhead of ~5 us) is the floor; the HBM fraction of this row is reported for completeness only")
        add(f"dvr.render[M={M}]", hip_time(lambda: dvr.render(sigma, origin, points, tindex, "l1")),
            2 * vol + N * M * 24 + cnt * 12,
            bound="step-parallel traversal, lane-per-step fp32 atomics (coalesced along a ray), not HBM")
        del out, em, live
    from vidar_amd.third_lib.chamferdist import knn_points
    rng = np.random.default_rng(0)
    a3, b3 = (torch.from_numpy(rng.uniform(-50, 50, (1, 30000, 3)).astype(np.float32)).to(dev) for _ in range(2))
    ms = hip_time(lambda: knn_points(a3, b3))
    add("chamferdist.knn_points_idx[30000x30000]", ms, 12 * 60000 + 12 * 30000,
        bound="fp32 VALU (9e8 pair evaluations, 8 flop each), not HBM",
        note=f"{9e8 / ms / 1e9:.2f} Tpairs/s = {9e8 * 8 / ms / 1e9:.1f} TFLOP/s fp32")
    from vidar_amd.synthetic import msda_operands, msda_operands_coherent
    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    # random reference points (no two queries share a line: the worst case for the caches) and, for the cross attention,
    # spatially coherent queries like the ones the model produces (neighbouring BEV queries project next to each other)
    for name, B, shapes, Nq, P, coherent in (("TSA", 2, [(200, 200)], 40000, 4, False), ("SCA", 6, fpn, 10000, 8, False),
                                             ("SCA, coherent queries", 6, fpn, 10000, 8, True)):
        value, sh, lsi, loc, w = (msda_operands_coherent(0, B, shapes, Nq, P=P, px=2.0, device=dev) if coherent
                                  else msda_operands(0, B, shapes, Nq, P=P, device=dev))
        L = len(shapes); Nv = value.shape[1]
        go = torch.randn(B, Nq, 256, device=dev)
        add(f"msda_fwd[{name}]", hip_time(lambda: _msda_forward(value, sh, lsi, loc, w)),
            msda_fwd_bytes(B, Nv, 8, 32, Nq, L, P),
            bound="L1 (TCP) bandwidth: 61 M corner lines x 128 B through the vector caches, TA busy 79 % "
                  "(profiles/r03_pmc_msda_sca); reported vs HBM")
        add(f"msda_bwd[{name}]", hip_time(lambda: _msda_backward(value, sh, lsi, loc, w, go)),
            msda_bwd_bytes(B, Nv, 8, 32, Nq, L, P))
    return rows


MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}       # MI355X_MICROARCH.md: dense fp32 / bf16 matrix-core peaks


def library_switches():
    """the A/B switches the measured build ran with (each setter returns the previous value: read = set + restore), so
    that two bench lines stay comparable when a default changes"""
    from vidar_amd._lib import lib
    L = lib()

    def peek(fn, probe):
        prev = fn(probe)
        fn(prev)
        return prev
    return {"dvr_traversal": peek(L.vidar_dvr_set_traversal, -1), "dvr_sort_min_waves": peek(L.vidar_dvr_set_sort_min_waves, 1024),
            "dvxlr_pad_mode": peek(L.vidar_dvxlr_set_pad_mode, 1), "gemm_variant": peek(L.vidar_gemm_set_variant, 0),
            "msda_item_order": peek(L.vidar_msda_set_item_order, 1), "dcn_variant": peek(L.vidar_dcn_set_variant, 1),
            "shortcut_accumulate": os.environ.get("VIDAR_SHORTCUT_ACCUMULATE", "1") != "0",
            "gradient_exchange": os.environ.get("VIDAR_DDP", "flat"), "fused_adamw": os.environ.get("VIDAR_FUSED_ADAMW", "1") != "0",
            "conv_offset": os.environ.get("VIDAR_CONV_OFFSET", "lib"), "share_subsample": os.environ.get("VIDAR_SHARE_SUBSAMPLE", "1") != "0"}


def gemm_rooflines(dev):
    """`roofline_gemm`: the hand-written MFMA GEMM (csrc/gemm_mfma.hip) at the two shapes that matter -- the attention
    value projection the north star assigns to MFMA ([6*30825, 256] x [256, 256], spatial_cross_attention.py:333-340)
    and the largest 1x1 convolution of the backbone (layer3 conv3: 24 x [1024, 256] x [256, 5800]) -- in both arithmetic
    modes, each against BOTH its rooflines: HBM (compulsory bytes / ms vs 8 TB/s) and the matrix cores (MFMA flops
    actually issued / ms vs the dense peak of the instruction: fp32 157 TF/s; bf16x3 issues 3 bf16 MFMAs per product
    -> 3 x 2MNK vs 2.5 PF/s), next to the library kernel on the same operands."""
    from vidar_amd import gemm as G
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *sh: torch.rand(*sh, device=dev, generator=g) * 2 - 1
    rows = []

    def add(name, flops, nbytes, fns):
        for mode, fn in fns.items():
            ms = hip_time(fn)
            mult, peak = (3.0, MFMA_PEAK_TFLOPS["bf16"]) if mode == "bf16x3" else (1.0, MFMA_PEAK_TFLOPS["f32"])
            rows.append(dict(kernel=name, mode=mode, avg_ms=round(ms, 4), bytes=int(nbytes), flops=flops,
                             hbm=dict(achieved=round(nbytes / ms / 1e6, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                      frac=round(nbytes / ms / 1e6 / HBM_PEAK_GBPS, 4)),
                             mfma=dict(achieved=round(mult * flops / ms / 1e9, 1), peak=peak, unit="TFLOP/s",
                                       frac=round(mult * flops / ms / 1e9 / peak, 4),
                                       note="library kernel: its own fp32 MFMA instructions" if mode == "lib" else None)))
    M, K, N = 6 * 30825, 256, 256
    x, w, b = rnd(M, K), rnd(N, K) * 0.1, rnd(N)
    add(f"value_proj fwd [{M},{K}]x[{K},{N}]", 2.0 * M * N * K, 4 * (M * K + M * N + N * K),
        {"lib": lambda: torch.addmm(b, x, w.t()), "f32": lambda: G.linear_forward(x, w, b, False, G.F32),
         "bf16x3": lambda: G.linear_forward(x, w, b, False, G.BF16X3)})
    Bn, Co, Ci, HW = 24, 1024, 256, 5800
    wc, xc = rnd(Co, Ci) * 0.05, rnd(Bn, Ci, HW)
    add(f"layer3 conv3 {Bn}x[{Co},{Ci}]x[{Ci},{HW}]", 2.0 * Bn * Co * Ci * HW, 4 * (Bn * Ci * HW + Bn * Co * HW + Co * Ci),
        {"lib": lambda: torch.bmm(wc.view(1, Co, Ci).expand(Bn, -1, -1), xc),
         "f32": lambda: G.conv_forward(wc, xc, precision=G.F32), "bf16x3": lambda: G.conv_forward(wc, xc, precision=G.BF16X3)})
    return rows


def timed_steps(step, steps, warmup, grouped, cuda, after_warmup=None, markers=None):
    """the contract's timing: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier +
    synchronize on both sides; -> seconds, MAX over ranks (device-agnostic: the gloo test drives it on CPU)"""
    def sync():
        if grouped:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    if after_warmup is not None:
        after_warmup()
    sync()
    if markers:
        markers(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if markers:
        markers(2)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    return elapsed


def allreduce_alone_ms(nbytes, dev, iters=5):
    """the collective by itself: `iters` all-reduces of an fp32 buffer of the step's gradient bytes on an otherwise idle
    device (HIP events around the batch), so that a record says how much of the step the exchange COULD cost when
    nothing hides it -- next to the step times of the exchange modes (`ddp_ab`) this tells overlap from overhead"""
    if not nbytes:
        return None
    buf = torch.zeros(max(1, nbytes // 4), device=dev)
    dist.all_reduce(buf)
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


def ddp_info(ddp, world):
    """what the gradient all-reduce moves per step (RCCL over xGMI): DDP's own bucket accounting.  With
    find_unused_parameters=False torch starts with ONE bucket and rebuilds the buckets after the first step in
    the order gradients became ready -- `rebuilt_bucket_bytes` is what overlaps with backward from step 2 on."""
    if hasattr(ddp, "logging_data"):                                 # vidar_amd.train.FlatAllReduce (the default)
        return dict(ddp.logging_data(), backend=dist.get_backend(), world_size=world, rccl_ranks=dist.get_world_size())
    if not hasattr(ddp, "_get_ddp_logging_data"):
        return None
    try:
        info = ddp._get_ddp_logging_data()
        ints = lambda key: [int(x) for x in str(info.get(key, "")).split(",") if x.strip()]
        sizes, rebuilt = ints("bucket_sizes"), ints("rebuilt_bucket_sizes")
        live = rebuilt or sizes
        return {"buckets": len(live), "bucket_bytes": live, "allreduce_bytes_per_step": sum(live),
                "initial_bucket_bytes": sizes, "rebuilt_bucket_bytes": rebuilt,
                "has_rebuilt_buckets": bool(info.get("has_rebuilt_buckets", 0)), "bucket_cap_mb": 100,
                "mode": "torch DistributedDataParallel (bucketed, overlapped by the reducer)",
                "backend": dist.get_backend(), "world_size": world, "rccl_ranks": dist.get_world_size()}
    except Exception as e:                                           # logging only, never fail the bench
        return {"error": str(e)[:100]}


def make_batch(cfg, args, rank, dev, spg=None):
    from vidar_amd.synthetic import fpn_features, make_sample
    spg = spg or args.samples_per_gpu
    samples = [make_sample(seed=100 + rank * spg + i, queue_length=cfg["queue_length"],
                           future_frames=cfg["future_frames"], rays_per_frame=args.rays_per_frame,
                           num_cams=cfg["num_cams"], img_hw=cfg["img_hw"]) for i in range(spg)]
    batch = dict(img_metas=[m for m, _ in samples], gt_points=[torch.from_numpy(g).to(dev) for _, g in samples])
    if args.no_backbone:
        batch["img_feats"] = fpn_features(200 + rank, cfg["queue_length"] + 1, num_cams=cfg["num_cams"],
                                          shapes=cfg["fpn_shapes"], device=dev, bs=spg)
    else:
        batch["img"] = torch.cat([synthetic_images(200 + rank * spg + i, cfg["queue_length"] + 1,
                                                   cfg["num_cams"], cfg["img_hw"], dev) for i in range(spg)])
    return batch


def run_config(name, args, rank, local, world, dev, steps, warmup, with_markers, spg=None, gemm_mode=None, tune=True,
               weights=None, ddp_mode=None):
    """build the model of one named config, time `steps` training steps -> dict(elapsed, ops, ddp, cfg, peak_mem_gb)"""
    from vidar_amd import gemm as G
    prev = G.set_mode(gemm_mode or G.mode())
    try:
        torch.cuda.reset_peak_memory_stats(dev)
        t0 = time.perf_counter()
        r = _run_config(name, args, rank, local, world, dev, steps, warmup, with_markers, spg, tune, weights or args.weights,
                        ddp_mode)
        if rank == 0:
            print(f"[bench] {name} spg={spg or args.samples_per_gpu} gemm={G.mode()} weights={r['weights']['mode']}: "
                  f"{r['elapsed'] / steps * 1e3:.1f} ms/step "
                  f"({time.perf_counter() - t0:.0f} s wall incl. build + warm-up)", file=sys.stderr, flush=True)
        r["gemm"] = G.mode()
        # the number next to the reference's only published figure for this path (README.md:143-148: ~63 GB for
        # vidar_1_8_nusc_3future, ~34 GB for its memory-efficient variant, per A100)
        r["peak_mem_gb"] = torch.cuda.max_memory_allocated(dev) / 1e9
        r["peak_reserved_gb"] = torch.cuda.max_memory_reserved(dev) / 1e9
        return r
    finally:
        G.set_mode(prev)


GEMM_DTYPE = {"lib": "f32", "auto": "f32", "f32": "f32",
              "bf16x3": "f32 storage, bf16x3 MFMA products (16-bit significand >= TF32), f32 accumulate"}


def _run_config(name, args, rank, local, world, dev, steps, warmup, with_markers, spg=None, tune=True, weights="init",
                ddp_mode=None):
    from vidar_amd import gemm_tuning
    from vidar_amd import train as T
    from vidar_amd import weights as W
    from vidar_amd._lib import TIMER
    from vidar_amd._lib import lib as _hip
    from vidar_amd.configs import get_config
    cfg = get_config(name, with_backbone=not args.no_backbone)
    torch.manual_seed(1234)                      # identical initial weights on every rank
    np.random.seed(1000 + rank)
    model = T.build_model(cfg).to(dev).train()
    batch = make_batch(cfg, args, rank, dev, spg)
    # the weight state is set BEFORE the data-parallel wrapper broadcasts rank 0's parameters: every rank then times the
    # same model (the calibration pass of `trained_like` sees each rank's own batch)
    wrep = W.prepare(model, batch, weights) if weights in W.MODES else W.prepare(model, batch, checkpoint=weights)
    ddp = T.wrap_ddp(model, local, mode=ddp_mode)
    opt = T.build_optimizer(model)
    grouped = dist.is_available() and dist.is_initialized()

    def after_warmup():
        # GEMM shapes were tuned (or loaded) during warm-up: freeze the selection, so no rank ever tunes inside
        # the timed region while the others wait at the all-reduce
        gemm_tuning.freeze()
        TIMER.reset()
        TIMER.enabled = True

    # only the main record tunes GEMM shapes it has not seen (the shipped solutions cover it); an extra config with new
    # shapes (a larger per-GPU batch) would spend minutes trying hundreds of library solutions per shape
    if tune:
        gemm_tuning.thaw()
    else:
        gemm_tuning.freeze()
    elapsed = timed_steps(lambda: T.train_step(ddp, opt, batch, cfg["grad_clip"]), steps, warmup, grouped, True,
                          after_warmup=after_warmup,
                          markers=(lambda k: _hip().vidar_marker(k, None)) if with_markers else None)
    TIMER.enabled = False
    ops = TIMER.summary() if rank == 0 else {}
    info = ddp_info(ddp, world) if grouped else None
    del batch, ddp, opt, model
    torch.cuda.empty_cache()
    if info and "allreduce_bytes_per_step" in info:
        info["allreduce_alone_ms"] = allreduce_alone_ms(info["allreduce_bytes_per_step"], dev)
    return dict(elapsed=elapsed, ops=ops, ddp=info, cfg=cfg, weights=wrep)


def main():
    args = parse()
    cmd = relaunch_command(args, sys.argv[1:])
    if cmd is not None:                  # `python bench.py --gpus N` called plainly: become N RCCL ranks
        print(f"[bench] --gpus {args.gpus} without a launcher: re-running as {' '.join(cmd[1:8])} ...", file=sys.stderr,
              flush=True)
        os.execv(cmd[0], cmd)
    if args.cpu_baseline_only:
        # never drive the host out of memory: cap this child's address space (a full-size CPU step took a box down)
        try:
            import resource
            total = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
            cap = min(total // 2, (1 << 40) if args.cpu_baseline_only == "full" else (64 << 30))
            resource.setrlimit(resource.RLIMIT_AS, (cap, cap))
        except (ImportError, ValueError, OSError):
            pass
        if args.cpu_baseline_only == "ops":
            print(json.dumps(cpu_baseline_ops(args.cpu_threads)), flush=True)
        else:
            print(json.dumps(cpu_baseline(args.config, args.cpu_threads, not args.no_backbone,
                                          reduced=args.cpu_baseline_only == "step")), flush=True)
        return
    from vidar_amd import gemm_tuning
    from vidar_amd import train as T

    rank, local, world = T.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    tuning = dict(enabled=False) if args.no_gemm_tuning else gemm_tuning.enable(results_file=args.tunableop_file, rank=rank)
    grouped = dist.is_available() and dist.is_initialized()
    spg = args.samples_per_gpu

    main_run = run_config(args.config, args, rank, local, world, dev, args.steps, args.warmup, with_markers=True,
                          gemm_mode=args.gemm)
    elapsed, ops, cfg = main_run["elapsed"], main_run["ops"], main_run["cfg"]
    # the gradient-exchange modes side by side on the headline config, whenever a process group exists: which of them is
    # fastest at N ranks is decided by THIS measurement (on one rank they only differ by host overhead)
    ddp_ab = []
    if grouped and not args.no_ddp_ab:
        main_mode = os.environ.get("VIDAR_DDP", "flat")
        ddp_ab.append({"mode": main_mode, "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps, "headline": True})
        for mode in T.DDP_MODES:
            if mode == main_mode:
                continue
            try:
                r = run_config(args.config, args, rank, local, world, dev, args.extra_steps, args.extra_warmup,
                               with_markers=False, gemm_mode=args.gemm, tune=False, ddp_mode=mode)
                ddp_ab.append({"mode": mode, "ms_per_step": r["elapsed"] / args.extra_steps * 1e3, "steps": args.extra_steps,
                               "ddp": r["ddp"]})
            except Exception as e:                                          # noqa: BLE001
                ddp_ab.append({"mode": mode, "error": f"{type(e).__name__}: {e}"[:200]})
                torch.cuda.empty_cache()
    extras = []
    t_extras = time.perf_counter()
    for entry in [c for c in args.extra_configs.split(",") if c]:
        rigor = entry.endswith("!")
        entry = entry.rstrip("!")
        xsteps, xwarm = (args.steps, args.warmup) if rigor else (args.extra_steps, args.extra_warmup)
        entry, _, xweights = entry.partition("+")
        xweights = xweights or args.weights
        head, _, xgemm = entry.partition(":")
        name, _, xs = head.partition("@")
        xspg = int(xs) if xs else spg
        xgemm = xgemm or None
        if (name == args.config and xspg == spg and (xgemm or main_run["gemm"]) == main_run["gemm"]
                and xweights == args.weights):
            continue
        # an extra config must never cost the main measurement its record: a failure (e.g. out of memory at a large
        # per-GPU batch) is agreed on across ranks and written into the entry; so is running out of the time budget
        err = None
        over = torch.tensor([1.0 if time.perf_counter() - t_extras > args.extra_budget_s else 0.0], device=dev)
        if grouped:
            dist.all_reduce(over, op=dist.ReduceOp.MAX)
        if float(over) > 0:
            extras.append({"config": name, "samples_per_gpu": xspg, "gemm": xgemm or main_run["gemm"], "weights": xweights,
                           "skipped": f"extra-config time budget ({args.extra_budget_s:.0f} s) spent"})
            continue
        try:
            r = run_config(name, args, rank, local, world, dev, xsteps, xwarm, with_markers=False,
                           spg=xspg, gemm_mode=xgemm, tune=False, weights=xweights)
        except Exception as e:                                              # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
            r = None
            torch.cuda.empty_cache()
        if grouped:
            flag = torch.tensor([1.0 if err else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag) > 0 and err is None:
                err = "failed on another rank"
        if err:
            extras.append({"config": name, "samples_per_gpu": xspg, "gemm": xgemm or main_run["gemm"], "weights": xweights,
                           "error": err})
            continue
        rec = {"config": name, "samples_per_gpu": xspg, "value": world * xspg * xsteps / r["elapsed"],
               "unit": "samples/s", "ms_per_step": r["elapsed"] / xsteps * 1e3, "steps": xsteps,
               "warmup": xwarm, "n_gpus": world, "global_batch": world * xspg,
               "cameras": r["cfg"]["num_cams"], "img_hw": list(r["cfg"]["img_hw"]), "gemm": r["gemm"],
               "dtype": GEMM_DTYPE[r["gemm"]], "weights": r["weights"]["mode"], "peak_mem_gb": round(r["peak_mem_gb"], 2),
               "peak_reserved_gb": round(r["peak_reserved_gb"], 2)}
        if r["ops"]:
            dn, dv = max(((k, v) for k, v in r["ops"].items() if not k.startswith(("dcn_", "affine_act", "stem_", "gemm_"))),
                         key=lambda kv: kv[1]["total_ms"])
            ach = dv["bytes_per_call"] / (dv["avg_ms"] * 1e-3) / 1e9
            rec["roofline"] = {"bound": "hbm", "kernel": dn, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBPS, "avg_ms": dv["avg_ms"],
                               "launches_per_step": dv["calls"] / xsteps}
        if r["ddp"]:
            rec["ddp"] = r["ddp"]
        extras.append(rec)

    if rank == 0:
        if args.op_table:
            for k, v in sorted(ops.items(), key=lambda kv: -kv[1]["total_ms"]):
                gb = v["bytes_per_call"] / v["avg_ms"] / 1e6 if v["avg_ms"] > 0 else 0
                unit = "GFLOP/s" if k.startswith("gemm_") else "GB/s"      # the gemm spans carry 2MNK, not bytes
                print(f"{k:28s} calls/step {v['calls'] / args.steps:6.1f}  avg {v['avg_ms']:8.3f} ms  "
                      f"total/step {v['total_ms'] / args.steps:8.2f} ms  alg {gb:8.1f} {unit}", file=sys.stderr)
        # `roofline`: the dominant kernel of the SURVEY 8(a) hot path (MSDA / latent render / ray march / chamfer);
        # the backbone's kernels (row f-1: dcn_*, affine_act_*) compete in `roofline_step_dominant`
        hot = {k: v for k, v in ops.items() if not k.startswith(("dcn_", "affine_act", "stem_", "gemm_"))} or ops
        dom_name, dom = max(hot.items(), key=lambda kv: kv[1]["total_ms"])
        achieved = dom["bytes_per_call"] / (dom["avg_ms"] * 1e-3) / 1e9
        all_name, all_dom = max(((k, v) for k, v in ops.items() if not k.startswith("gemm_")), key=lambda kv: kv[1]["total_ms"])
        all_ach = all_dom["bytes_per_call"] / (all_dom["avg_ms"] * 1e-3) / 1e9
        hip_ms = sum(v["total_ms"] for v in ops.values()) / args.steps
        out = {
            "metric": "train samples/sec (6-cam->BEV step)", "value": world * spg * args.steps / elapsed,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": GEMM_DTYPE[main_run["gemm"]], "data": "synthetic", "gemm": main_run["gemm"],
            "weights": main_run["weights"],
            "peak_mem_gb": round(main_run["peak_mem_gb"], 2), "peak_reserved_gb": round(main_run["peak_reserved_gb"], 2),
            "config": {"workload": (f"{args.config}: images [1,5,{cfg['num_cams']},3,{cfg['img_hw'][0]}x"
                                    f"{cfg['img_hw'][1]}] -> ResNet101-DCNv2 + FPN -> " if not args.no_backbone
                                    else f"{args.config} (no image backbone): FPN pyramids -> ")
                                   + "5x BEV encode (6 layers TSA+SCA, LatentRendering, bev 200x200) -> head -> "
                                     "ray CE + gumbel render + chamfer -> backward -> clip -> AdamW",
                       "global_batch": world * spg, "rays_per_frame": args.rays_per_frame,
                       "parallelism": f"dp{world}", "weights": main_run["weights"]["mode"]},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(dom_name)[0], "traffic_source": pmc_traffic(dom_name)[1],
                         "avg_ms": dom["avg_ms"], "launches_per_step": dom["calls"] / args.steps,
                         "ms_per_step": dom["total_ms"] / args.steps, "hip_ops_ms_per_step": hip_ms},
            "gemm_tuning": dict(tuning, solutions=gemm_tuning.count_results() if tuning.get("enabled") else 0,
                                frozen_after_warmup=bool(tuning.get("enabled"))),
            "roofline_step_dominant": {"bound": "hbm", "kernel": all_name, "achieved": all_ach, "peak": HBM_PEAK_GBPS,
                                       "unit": "GB/s", "frac": all_ach / HBM_PEAK_GBPS, "avg_ms": all_dom["avg_ms"],
                                       "launches_per_step": all_dom["calls"] / args.steps,
                                       "ms_per_step": all_dom["total_ms"] / args.steps},
        }
        try:
            out["switches"] = library_switches()
        except Exception as e:                                              # noqa: BLE001  (reporting only)
            out["switches"] = {"error": str(e)[:100]}
        if extras:
            out["configs"] = extras
        if main_run["ddp"]:
            out["ddp"] = main_run["ddp"]
        if ddp_ab:
            out["ddp_ab"] = ddp_ab
        # the two auxiliary legs run after the timed region; a failure in one of them is reported in the line
        # (and on stderr) instead of costing the measured throughput its record
        if world == 1 and not args.no_kernel_rooflines and not grouped:
            print("[bench] kernel rooflines ...", file=sys.stderr, flush=True)
            try:
                out["roofline_kernels"] = kernel_rooflines(dev)
            except Exception as e:                                          # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["roofline_kernels_error"] = f"{type(e).__name__}: {e}"
            try:
                out["roofline_gemm"] = gemm_rooflines(dev)
            except Exception as e:                                          # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["roofline_gemm_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not args.no_cpu_baseline and not grouped:
            print("[bench] cpu baseline (child process) ...", file=sys.stderr, flush=True)
            try:
                out["cpu_baseline"] = cpu_baseline_record(args, ops, args.steps, out.get("roofline_kernels"))
            except Exception as e:                                          # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
