"""bench.py -- train samples/s of ViDAR's hot path (6-cam FPN features -> BEV encode ->
latent render -> occupancy head -> ray-march / chamfer losses -> backward -> AdamW) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU, DDP gradient all-reduce over RCCL.  Prints ONE JSON line on rank 0 with the
driver contract fields plus `roofline` (dominant HIP kernel, timed live with HIP events on the
launch stream inside the timed region) and `cpu_baseline` (the CPU oracle port of the same step on a
bounded sample, rank 0 / N=1 only).  A "sample" = one 5-frame x 6-camera sequence (global batch =
number of GPUs, the reference asserts 1 sample per GPU: detectors/vidar.py:306)."""
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
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
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


def synthetic_images(seed, T, num_cams, hw, device, scale=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(1, T, num_cams, 3, hw[0] // scale, hw[1] // scale, generator=g).to(device)


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


def _peak_rss_gb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def cpu_baseline_subprocess(args):
    """Run the CPU leg in a child with a hard wall-clock limit so it can never stall the bench."""
    import subprocess
    base = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", "--config", args.config,
            "--cpu-threads", str(args.cpu_threads)] + (["--no-backbone"] if args.no_backbone else [])
    env = dict(os.environ, OMP_NUM_THREADS=str(args.cpu_threads), MKL_NUM_THREADS=str(args.cpu_threads))
    note = ""
    for extra, limit in ((["--cpu-baseline-full"], args.cpu_baseline_timeout), ([], 240)):
        if extra and not args.cpu_baseline_full:
            continue
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=limit, env=env)
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{"):
                    out = json.loads(line)
                    if note:
                        out["sample"] += " (" + note + ")"
                    return out
            note += "cpu leg produced no result: " + r.stderr.strip()[-200:] + "; "
        except subprocess.TimeoutExpired:
            note += f"cpu leg{' (full size)' if extra else ' (bounded sample)'} exceeded {limit} s and was stopped; "
    return dict(value=None, unit="samples/s", cores=args.cpu_threads, kind="port", sample=note)


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
    for T_, rpf in ((1, 30000), (5, 30000)):
        sigma, origin, points, tindex = map(t, ray_set(seed=0, N=1, T=T_, rays_per_frame=rpf))
        N, M = tindex.shape
        vol = sigma.numel() * 4
        out = dvxlr.render(sigma, origin, points, tindex)
        cnt = int(((out[3] != 0).any(-1)).sum())             # traversed voxels (dvr bytes depend on it)
        add(f"dvxlr.render[M={M}]", hip_time(lambda: dvxlr.render(sigma, origin, points, tindex)),
            vol + N * M * 16 + N * M * 4 * (2 + 1026 * 4), note="16.4 KB/ray of API-mandated padded rows")
        em = out[2] * 0.5
        add(f"dvxlr.get_grad_sigma[M={M}]", hip_time(lambda: dvxlr.get_grad_sigma(em, out[3], tindex, sigma)),
            N * M * 1026 * 16 + 2 * vol)
        add(f"dvxlr_v2.render_v2[M={M}]", hip_time(lambda: dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma)),
            2 * vol + N * M * 16 + N * M * 4 * (2 + 1026 * 6))
        add(f"dvr.render_forward[M={M}]",
            hip_time(lambda: dvr.render_forward(sigma, origin, points, tindex, [T_, 16, 200, 200], "train")),
            vol + N * M * 24 + cnt * 4, bound="fp64 issue (sequential DDA per lane), not HBM")
        add(f"dvr.render[M={M}]", hip_time(lambda: dvr.render(sigma, origin, points, tindex, "l1")),
            2 * vol + N * M * 24 + cnt * 12, bound="fp64 issue + per-lane atomics, not HBM")
        del out, em
    from vidar_amd.synthetic import msda_operands
    fpn = [(116, 200), (58, 100), (29, 50), (15, 25)]
    for name, B, shapes, Nq, P in (("TSA", 2, [(200, 200)], 40000, 4), ("SCA", 6, fpn, 10000, 8)):
        value, sh, lsi, loc, w = msda_operands(0, B, shapes, Nq, P=P, device=dev)
        L = len(shapes); Nv = value.shape[1]
        go = torch.randn(B, Nq, 256, device=dev)
        add(f"msda_fwd[{name}]", hip_time(lambda: _msda_forward(value, sh, lsi, loc, w)),
            msda_fwd_bytes(B, Nv, 8, 32, Nq, L, P), bound="L1/TA line rate (61 M corner lines), reported vs HBM")
        add(f"msda_bwd[{name}]", hip_time(lambda: _msda_backward(value, sh, lsi, loc, w, go)),
            msda_bwd_bytes(B, Nv, 8, 32, Nq, L, P))
    return rows


def main():
    args = parse()
    if args.cpu_baseline_only:
        # never drive the host out of memory: cap this child's address space (a full-size CPU step took a box down)
        try:
            import resource
            total = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
            cap = min(total // 2, (1 << 40) if args.cpu_baseline_full else (48 << 30))
            resource.setrlimit(resource.RLIMIT_AS, (cap, cap))
        except (ImportError, ValueError, OSError):
            pass
        print(json.dumps(cpu_baseline(args.config, args.cpu_threads, not args.no_backbone,
                                      reduced=not args.cpu_baseline_full)), flush=True)
        return
    from vidar_amd import train as T
    from vidar_amd._lib import TIMER
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import fpn_features, make_sample

    rank, local, world = T.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = get_config(args.config, with_backbone=not args.no_backbone)
    from vidar_amd import gemm_tuning
    tuning = dict(enabled=False) if args.no_gemm_tuning else gemm_tuning.enable(results_file=args.tunableop_file, rank=rank)
    torch.manual_seed(1234)                      # identical initial weights on every rank
    np.random.seed(1000 + rank)
    model = T.build_model(cfg).to(dev).train()
    ddp = T.wrap_ddp(model, local)
    opt = T.build_optimizer(model)
    spg = args.samples_per_gpu
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

    grouped = dist.is_available() and dist.is_initialized()

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        T.train_step(ddp, opt, batch, cfg["grad_clip"])
    from vidar_amd._lib import lib as _hip
    TIMER.reset()
    TIMER.enabled = True
    sync()
    _hip().vidar_marker(1, None)                 # delimits the timed region in a rocprofv3 trace
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T.train_step(ddp, opt, batch, cfg["grad_clip"])
    sync()
    elapsed = time.perf_counter() - t0
    _hip().vidar_marker(2, None)
    TIMER.enabled = False
    if grouped:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        ops = TIMER.summary()
        if args.op_table:
            for k, v in sorted(ops.items(), key=lambda kv: -kv[1]["total_ms"]):
                gb = v["bytes_per_call"] / v["avg_ms"] / 1e6 if v["avg_ms"] > 0 else 0
                print(f"{k:28s} calls/step {v['calls'] / args.steps:6.1f}  avg {v['avg_ms']:8.3f} ms  "
                      f"total/step {v['total_ms'] / args.steps:8.2f} ms  alg {gb:8.1f} GB/s", file=sys.stderr)
        # `roofline`: the dominant kernel of the SURVEY 8(a) hot path (MSDA / latent render / ray march / chamfer);
        # the backbone's kernels (row f-1: dcn_*, affine_act_*) compete in `roofline_step_dominant`
        hot = {k: v for k, v in ops.items() if not k.startswith(("dcn_", "affine_act"))} or ops
        dom_name, dom = max(hot.items(), key=lambda kv: kv[1]["total_ms"])
        achieved = dom["bytes_per_call"] / (dom["avg_ms"] * 1e-3) / 1e9
        all_name, all_dom = max(ops.items(), key=lambda kv: kv[1]["total_ms"])
        all_ach = all_dom["bytes_per_call"] / (all_dom["avg_ms"] * 1e-3) / 1e9
        hip_ms = sum(v["total_ms"] for v in ops.values()) / args.steps
        out = {
            "metric": "train samples/sec (6-cam->BEV step)", "value": world * spg * args.steps / elapsed,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.config}: images [1,5,{cfg['num_cams']},3,{cfg['img_hw'][0]}x"
                                    f"{cfg['img_hw'][1]}] -> ResNet101-DCNv2 + FPN -> " if not args.no_backbone
                                    else f"{args.config} (no image backbone): FPN pyramids -> ")
                                   + "5x BEV encode (6 layers TSA+SCA, LatentRendering, bev 200x200) -> head -> "
                                     "ray CE + gumbel render + chamfer -> backward -> clip -> AdamW",
                       "global_batch": world * spg, "rays_per_frame": args.rays_per_frame,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(dom_name)[0], "traffic_source": pmc_traffic(dom_name)[1],
                         "avg_ms": dom["avg_ms"], "launches_per_step": dom["calls"] / args.steps,
                         "ms_per_step": dom["total_ms"] / args.steps, "hip_ops_ms_per_step": hip_ms},
            "gemm_tuning": dict(tuning, solutions=gemm_tuning.count_results() if tuning.get("enabled") else 0),
            "roofline_step_dominant": {"bound": "hbm", "kernel": all_name, "achieved": all_ach, "peak": HBM_PEAK_GBPS,
                                       "unit": "GB/s", "frac": all_ach / HBM_PEAK_GBPS, "avg_ms": all_dom["avg_ms"],
                                       "launches_per_step": all_dom["calls"] / args.steps,
                                       "ms_per_step": all_dom["total_ms"] / args.steps},
        }
        if grouped and hasattr(ddp, "_get_ddp_logging_data"):
            # what the gradient all-reduce moves per step (RCCL over xGMI): DDP's own bucket accounting
            try:
                info = ddp._get_ddp_logging_data()
                sizes = [int(x) for x in str(info.get("bucket_sizes", "")).split(",") if x.strip()]
                out["ddp"] = {"buckets": len(sizes), "bucket_bytes": sizes, "allreduce_bytes_per_step": sum(sizes),
                              "bucket_cap_mb": 100, "backend": dist.get_backend(), "world_size": world}
            except Exception as e:                                   # logging only, never fail the bench
                out["ddp"] = {"error": str(e)[:100]}
        if world == 1 and not args.no_kernel_rooflines and not grouped:
            del batch, ddp, opt, model
            torch.cuda.empty_cache()
            out["roofline_kernels"] = kernel_rooflines(dev)
        if world == 1 and not args.no_cpu_baseline and not grouped:
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
            out["cpu_baseline"]["host_cores"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
