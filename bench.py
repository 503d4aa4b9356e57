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
# doubled: the x2 rule of the microarch guide is calibrated for wide coalesced streams only).
# Source: profiles/r01_pmc_{FETCH,WRITE}_SIZE_kbench_msda_lr.csv (tools/pmc_traffic.sh, SCA shape
# B=6, Nq=10^4, L=4, P=8 / TSA shape B=2, Nq=4*10^4).  Atomic read-modify-writes show up as writes.
PMC_TRAFFIC = {"msda_bwd[L=4,P=8]": (1392835.0 + 5508074.1) * 1024,
               # TSA row = 3*mean - min(Pred) - max(SCA) of the 39 equally split dispatches
               "msda_bwd[L=1,P=4]": (457348.7 + 920731.2) * 1024}


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
    ap.add_argument("--cpu-baseline-timeout", type=int, default=240)
    ap.add_argument("--op-table", action="store_true", help="print the per-op timing table to stderr")
    return ap.parse_args()


def synthetic_images(seed, T, num_cams, hw, device, scale=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(1, T, num_cams, 3, hw[0] // scale, hw[1] // scale, generator=g).to(device)


def cpu_baseline(config, threads, with_backbone=True):
    """The oracle port of the same training step on host cores, bounded sample: BEV 50x50 (1/16 of
    the 40 000 queries), FPN pyramid of a quarter-resolution input, rays_per_frame/16 GT rays."""
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import fpn_features, make_sample
    torch.set_num_threads(threads)
    cfg = get_config(config, bev_h=50, bev_w=50, with_backbone=with_backbone)
    torch.manual_seed(0); np.random.seed(0)
    model = T.build_model(cfg).train()
    opt = T.build_optimizer(model)
    metas, gt = make_sample(0, rays_per_frame=30000 // 16, future_frames=cfg["future_frames"],
                            num_cams=cfg["num_cams"], img_hw=cfg["img_hw"])
    if with_backbone:
        qhw = (cfg["img_hw"][0] // 4, cfg["img_hw"][1] // 4)
        for m in metas:
            m["img_shape"] = [(qhw[0], qhw[1], 3)] * cfg["num_cams"]
            k = np.diag([0.25, 0.25, 1.0, 1.0])
            m["lidar2img"] = [k @ a for a in m["lidar2img"]]
        batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt)],
                     img=synthetic_images(0, 5, cfg["num_cams"], cfg["img_hw"], "cpu", scale=4))
    else:
        shapes = [((h + 3) // 4, (w + 3) // 4) for h, w in cfg["fpn_shapes"]]
        feats = fpn_features(0, 5, num_cams=cfg["num_cams"], shapes=shapes)
        batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt)], img_feats=feats)
    with cpu_ops.patched():
        T.train_step(model, opt, batch)             # warm-up
        t0 = time.perf_counter()
        n = 1
        for _ in range(n):
            T.train_step(model, opt, batch)
        dt = (time.perf_counter() - t0) / n
    scale = 16.0
    return dict(value=1.0 / (dt * scale), unit="samples/s", cores=threads, kind="port",
                sample=f"oracle port of the step ({'with' if with_backbone else 'without'} backbone) at BEV "
                       f"50x50, 1/4-res images (1/16 of the pixels), 1875 rays/frame: "
                       f"{dt:.2f} s/step measured, x{scale:.0f} work -> full-size estimate")


def cpu_baseline_subprocess(args):
    """Run the CPU leg in a child with a hard wall-clock limit so it can never stall the bench."""
    import subprocess
    cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", "--config", args.config,
           "--cpu-threads", str(args.cpu_threads)] + (["--no-backbone"] if args.no_backbone else [])
    env = dict(os.environ, OMP_NUM_THREADS=str(args.cpu_threads), MKL_NUM_THREADS=str(args.cpu_threads))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_baseline_timeout, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        note = "cpu leg produced no result: " + r.stderr.strip()[-200:]
    except subprocess.TimeoutExpired:
        note = f"cpu leg exceeded {args.cpu_baseline_timeout} s and was stopped"
    return dict(value=None, unit="samples/s", cores=args.cpu_threads, kind="port", sample=note)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.config, args.cpu_threads, not args.no_backbone)), flush=True)
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
        dom_name, dom = max(ops.items(), key=lambda kv: kv[1]["total_ms"])
        achieved = dom["bytes_per_call"] / (dom["avg_ms"] * 1e-3) / 1e9
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
                         "traffic": PMC_TRAFFIC.get(dom_name),
                         "avg_ms": dom["avg_ms"], "launches_per_step": dom["calls"] / args.steps,
                         "hip_ops_ms_per_step": hip_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
            out["cpu_baseline"]["host_cores"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
