"""Which library GEMMs / convolutions does one training step issue, with which shapes, how often?

    python tools/op_shapes.py [--config vidar_1_8_nusc_1future] [--out gpurun_out/op_shapes.txt]

The kernel trace names library kernels by their tiling (Cijk_..._MT64x64x32...), not by the operation that asked for
them.  This runs ONE step under torch.profiler with the CPU activity only (no device tracer is started: nothing here
can hang the GPU), with input shapes recorded, and lists every aten matrix product / convolution -- forward and the
autograd engine's backward nodes -- grouped by input shapes, with the FLOPs of the shape and its call count, so that a
kernel of the trace can be matched by (calls per step, duration): e.g. the trace's largest library row, 24 x 1.17 ms of
Cijk_..._MT64x64x32, is the DCN column product of the 24 no-grad history images (bmm [24,256,2304] x [24,2304,5800],
164 GFLOP -> 141 TFLOP/s), not a badly tiled small GEMM."""
import argparse
import os
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WANTED = ("aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm", "aten::convolution_backward", "aten::_convolution",
          "aten::miopen_convolution", "aten::addbmm", "aten::mv", "aten::_scaled_mm", "aten::linear")


def flops(name, shapes):
    try:
        if name in ("aten::mm",):
            (m, k), (_, n) = shapes[0], shapes[1]
            return 2 * m * k * n
        if name == "aten::addmm":
            (m, k), (_, n) = shapes[1], shapes[2]
            return 2 * m * k * n
        if name in ("aten::bmm",):
            (b, m, k), (_, _, n) = shapes[0], shapes[1]
            return 2 * b * m * k * n
        if name == "aten::baddbmm":
            (b, m, k), (_, _, n) = shapes[1], shapes[2]
            return 2 * b * m * k * n
    except Exception:                                                       # noqa: BLE001
        return 0
    return 0


EAGER = ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::sub", "aten::div", "aten::copy_", "aten::cat", "aten::sum",
         "aten::clone", "aten::contiguous", "aten::index", "aten::index_select", "aten::gather", "aten::where", "aten::sigmoid",
         "aten::softmax", "aten::_softmax", "aten::relu", "aten::fill_", "aten::zero_", "aten::stack", "aten::masked_fill",
         "aten::masked_fill_", "aten::native_dropout", "aten::native_layer_norm", "aten::mean", "aten::sqrt", "aten::exp",
         "aten::addcmul", "aten::addcmul_", "aten::lerp_", "aten::_foreach_add_", "aten::new_zeros", "aten::zeros", "aten::zeros_like",
         "aten::index_put_", "aten::_index_put_impl_", "aten::scatter_add_", "aten::sigmoid_backward", "aten::_softmax_backward_data",
         "aten::native_dropout_backward", "aten::threshold_backward", "aten::permute", "aten::transpose")


def eager_table(prof, args):
    rows = defaultdict(int)
    for ev in prof.events():
        if ev.name in EAGER and ev.name not in ("aten::permute", "aten::transpose"):
            shapes = tuple(tuple(s) for s in (ev.input_shapes or []) if isinstance(s, (list, tuple)) and len(s))
            rows[(ev.name, shapes)] += 1

    def nbytes(shapes):
        return 4 * sum(int(np.prod(s)) for s in shapes)
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    lines = [f"# {args.config}: elementwise / copy / reduce aten ops of ONE training step (CPU activity; nested ops appear under both names)",
             f"{'calls':>6} {'MB/call':>9} {'MB total':>10}  op  input shapes"]
    for (name, shapes), n in sorted(rows.items(), key=lambda kv: -nbytes(kv[0][1]) * kv[1]):
        lines.append(f"{n:6d} {nbytes(shapes) / 1e6:9.2f} {n * nbytes(shapes) / 1e6:10.1f}  {name}  {[list(s) for s in shapes]}")
    out.write_text("\n".join(lines) + "\n")
    print("\n".join(lines[:90]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vidar_1_8_nusc_1future")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/op_shapes.txt")
    ap.add_argument("--eager", action="store_true", help="list the elementwise / copy / cat / reduce aten ops instead (the step's "
                    "eager tail), ranked by calls x operand bytes")
    args = ap.parse_args()
    import bench
    from vidar_amd import gemm_tuning
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    gemm_tuning.enable(rank=0, results_file=f"/tmp/op_shapes_tunableop_{os.getpid()}.csv")
    cfg = get_config(args.config, with_backbone=True)
    bargs = argparse.Namespace(samples_per_gpu=1, rays_per_frame=30000, no_backbone=False)
    torch.manual_seed(1234); np.random.seed(1000)
    model = T.build_model(cfg).to(dev).train()
    opt = T.build_optimizer(model)
    batch = bench.make_batch(cfg, bargs, 0, dev)
    for _ in range(args.warmup):
        T.train_step(model, opt, batch, cfg["grad_clip"])
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        T.train_step(model, opt, batch, cfg["grad_clip"])
        torch.cuda.synchronize()
    rows = defaultdict(int)
    if args.eager:
        return eager_table(prof, args)
    for ev in prof.events():
        if ev.name in WANTED:
            shapes = tuple(tuple(s) for s in (ev.input_shapes or []) if isinstance(s, (list, tuple)))
            rows[(ev.name, shapes)] += 1
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    lines = [f"# {args.config}: aten matrix products / convolutions of ONE training step (torch.profiler, CPU activity, shapes)",
             f"{'calls':>6} {'GFLOP':>9}  op  input shapes"]
    for (name, shapes), n in sorted(rows.items(), key=lambda kv: -flops(kv[0][0], kv[0][1]) * kv[1]):
        lines.append(f"{n:6d} {flops(name, shapes) / 1e9:9.2f}  {name}  {[list(s) for s in shapes if s]}")
    out.write_text("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
