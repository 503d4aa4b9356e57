"""Which torch ops own the eager kernels of a step?  torch.profiler over two training steps of the bench workload,
grouped by operator (self device time), the library's own ops and the GEMM / convolution ops listed separately.
    python tools/eager_ops.py [--config vidar_1_8_nusc_1future] [--no-backbone] > gpurun_out/eager_ops.txt"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vidar_1_8_nusc_1future")
    ap.add_argument("--no-backbone", action="store_true")
    a = ap.parse_args()
    import bench
    from vidar_amd import gemm_tuning
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    args = argparse.Namespace(samples_per_gpu=1, rays_per_frame=30000, no_backbone=a.no_backbone)
    dev = torch.device("cuda", 0)
    gemm_tuning.enable()
    cfg = get_config(a.config, with_backbone=not a.no_backbone)
    torch.manual_seed(1234); np.random.seed(1000)
    model = T.build_model(cfg).to(dev).train()
    opt = T.build_optimizer(model)
    batch = bench.make_batch(cfg, args, 0, dev)
    for _ in range(3):
        T.train_step(model, opt, batch, cfg["grad_clip"])
    gemm_tuning.freeze()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(steps):
            T.train_step(model, opt, batch, cfg["grad_clip"])
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = e.self_cuda_time_total
        if dt > 0:
            rows.append((dt / steps / 1e3, e.count / steps, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    print(f"{'ms/step':>9s} {'calls':>7s}  op  input shapes")
    for ms, n, key, shapes in rows[:60]:
        print(f"{ms:9.3f} {n:7.1f}  {key[:46]:46s} {shapes}")
    skip = ("aten::bmm", "aten::mm", "aten::addmm", "aten::miopen", "aten::convolution", "aten::_convolution")
    eager = [r for r in rows if r[2].startswith("aten::") and not r[2].startswith(skip)]
    print(f"\n# torch eager operators only: {sum(r[0] for r in eager):.2f} ms/step, {sum(r[1] for r in eager):.0f} calls/step")
    for ms, n, key, shapes in eager[:120]:
        print(f"{ms:9.3f} {n:7.1f}  {key[:34]:34s} {shapes}")


if __name__ == "__main__":
    main()
