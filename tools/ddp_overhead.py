"""Where do the +8 ms of a 1-rank RCCL run come from (357 vs 349 ms/step, profiles/r04_bench_ddp_1rank_rccl.json)?

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
        tools/ddp_overhead.py [--steps 10]

Times the same training step (same weights, same batch) five ways on ONE rank of a real RCCL group:
  plain      the bare module (what `bench.py --gpus 1` times)
  flat       vidar_amd.train.FlatAllReduce: one flat all-reduce after backward (the default of wrap_ddp)
  ddp        torch's DistributedDataParallel as wrap_ddp builds it under VIDAR_DDP=torch
  no_sync    that wrapper inside `no_sync()`: every autograd hook still runs, no bucket is reduced
  collective the bucket traffic alone: dist.all_reduce of tensors of the rebuilt bucket sizes, no model
so that `ddp - no_sync` = what the all-reduce launches cost and `no_sync - plain` = the reducer's host-side
bookkeeping (one autograd hook per parameter) on a step whose backward is launch-bound in places."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["VIDAR_FORCE_DDP"] = "1"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="vidar_1_8_nusc_1future")
    args = ap.parse_args()
    import bench
    from vidar_amd import gemm_tuning
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    rank, local, world = T.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    gemm_tuning.enable(rank=rank)
    cfg = get_config(args.config, with_backbone=True)
    bargs = argparse.Namespace(samples_per_gpu=1, rays_per_frame=30000, no_backbone=False)
    out = {}

    def run(tag, wrap, ctx=None):
        torch.manual_seed(1234); np.random.seed(1000)
        model = T.build_model(cfg).to(dev).train()
        m = wrap(model)
        opt = T.build_optimizer(model)
        batch = bench.make_batch(cfg, bargs, rank, dev)

        def step():
            if ctx is None:
                T.train_step(m, opt, batch, cfg["grad_clip"])
            else:
                with ctx(m):
                    T.train_step(m, opt, batch, cfg["grad_clip"])
        for _ in range(args.warmup):
            step()
        gemm_tuning.freeze()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        out[tag] = round((time.perf_counter() - t0) / args.steps * 1e3, 2)
        info = bench.ddp_info(m, world) if hasattr(m, "_get_ddp_logging_data") else None
        if hasattr(m, "logging_data"):
            out["flat_allreduce_bytes"] = m.logging_data()["allreduce_bytes_per_step"]
        del m, model, opt, batch
        torch.cuda.empty_cache()
        return info

    run("plain_ms", lambda mod: mod)
    os.environ["VIDAR_DDP"] = "flat"
    run("flat_allreduce_ms", lambda mod: T.wrap_ddp(mod, local))
    os.environ["VIDAR_DDP"] = "torch"
    info = run("ddp_ms", lambda mod: T.wrap_ddp(mod, local))
    run("ddp_no_sync_ms", lambda mod: T.wrap_ddp(mod, local), ctx=lambda m: m.no_sync())
    sizes = (info or {}).get("bucket_bytes") or [106 << 20, 144 << 20]
    bufs = [torch.zeros(max(1, b // 4), device=dev) for b in sizes]
    for _ in range(3):
        for b in bufs:
            dist.all_reduce(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        for b in bufs:
            b.div_(world); dist.all_reduce(b)
    torch.cuda.synchronize()
    out["collective_alone_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    out["bucket_bytes"] = sizes
    out["parameters_with_grad"] = sum(1 for p in T.build_model(cfg).parameters() if p.requires_grad)
    out["reducer_hooks_ms"] = round(out["ddp_no_sync_ms"] - out["plain_ms"], 2)
    out["allreduce_launches_ms"] = round(out["ddp_ms"] - out["ddp_no_sync_ms"], 2)
    print(json.dumps(out), flush=True)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
