"""Run the reference-pinned golden comparisons (tests/test_{transformer,head_v1,detector}_golden_cpu.py)
with the HIP kernels instead of the CPU oracle: the plugin on cuda:0 against the vectors the
reference's own Python produced.  Reports max abs / relative errors per quantity, e.g. the chamfer
distance per future frame of `ViDAR.forward_test` next to the reference's value (BASELINE:
"CD@1s within 1e-3 m").

Not part of the pytest suite yet: the golden models use widths the kernels were not exercised at on
hardware (embed 64 = 2 heads x 32 channels, 2 cameras, 2 pyramid levels, BEV 12x12) -- run this first
on a GPU box, then promote it to tests/ (-m gpu).      python tools/golden_gpu_check.py"""
import copy
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
GOLD = ROOT / "tests" / "golden"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


def encoder():
    from vidar_amd.plugin.registry import build_transformer
    g = np.load(GOLD / "transformer_encoder_small.npz", allow_pickle=False)
    model = build_transformer(json.loads(str(g["cfg_json"])))
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}, strict=True)
    model.cuda().eval()
    meta = dict(can_bus=g["can_bus"], lidar2global_rotation=g["lidar2global_rotation"],
                lidar2img=[m for m in g["lidar2img"]], img_shape=[tuple(int(v) for v in s) for s in g["img_shape"]])
    feats = [torch.from_numpy(g["feats0"]).cuda(), torch.from_numpy(g["feats1"]).cuda()]
    q = torch.from_numpy(g["bev_queries"]).cuda()
    B = int(round(q.shape[0] ** 0.5))
    kw = dict(grid_length=(102.4 / B, 102.4 / B), bev_pos=torch.from_numpy(g["bev_pos"]).cuda(), img_metas=[meta])
    with torch.no_grad():
        for key, prev in (("out_no_prev", None), ("out_prev", torch.from_numpy(g["prev_bev"]).cuda())):
            out = model.get_bev_features(feats, q, B, B, prev_bev=prev, **kw)
            print(f"encoder {key}: rel-L2 {rel(out.cpu().numpy(), g[key]):.2e}", flush=True)


def detector():
    import torch.nn.functional as F
    from vidar_amd.plugin.registry import build_detector
    from vidar_amd.synthetic import make_sample
    g = np.load(GOLD / "detector_small.npz", allow_pickle=False)
    model = build_detector(json.loads(str(g["cfg_json"])))
    model.init_weights()
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    sd["pts_bbox_head.code_weights"] = model.state_dict()["pts_bbox_head.code_weights"]
    model.load_state_dict(sd, strict=True)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.cuda()
    metas, gt = make_sample(4, queue_length=2, future_frames=3, rays_per_frame=50, num_cams=2)
    img = torch.randn(1, 3, 2, 3, 24, 40, generator=torch.Generator().manual_seed(8))
    gp = torch.Generator().manual_seed(21)
    shapes = [(12, 20), (6, 10)]
    proj = [torch.randn(64, 3, generator=gp) for _ in shapes]
    x = img.reshape(-1, *img.shape[-3:])
    feats = [torch.einsum("dc,nchw->ndhw", p, F.adaptive_avg_pool2d(x, s)).view(1, 3, 2, 64, *s).cuda()
             for s, p in zip(shapes, proj)]
    batch = dict(img_metas=[copy.deepcopy(metas)], gt_points=[torch.from_numpy(gt).cuda()], img_feats=feats)
    with torch.no_grad():
        res = model(return_loss=False, **batch)[0]
    for k, want in zip(g["test_keys"], g["test_values"]):
        r = res[str(k)]
        print(f"forward_test {k}: CD {r['chamfer_distance']:.5f} (ref {want[1]:.5f}, delta {r['chamfer_distance'] - want[1]:+.2e})"
              f"  L1 {r['l1_error']:.5f} (ref {want[2]:.5f})  AbsRel {r['absrel_error']:.5f} (ref {want[3]:.5f})", flush=True)
    noise = [(-torch.empty(*[int(v) for v in s]).exponential_(generator=torch.Generator().manual_seed(int(seed))).log())[0].cuda()
             for seed, s in zip(g["noise_seeds"], g["noise_shapes"])]
    calls = iter(noise)
    model.future_pred_head.gumbel_noise_fn = lambda R, K: next(calls)
    model.train()
    losses = model(return_loss=True, **copy.deepcopy(batch))
    for name, want in zip(g["loss_names"], g["loss_values"]):
        got = float(losses[str(name)])
        print(f"forward_train {name}: {got:.6f} (ref {want:.6f}, rel {abs(got - want) / max(abs(want), 1e-12):.2e})", flush=True)
    params = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    grads = torch.autograd.grad(sum(losses.values()), [params[n] for n in names])
    for i, (n, gr) in enumerate(zip(names, grads)):
        print(f"grad {n}: rel-L2 {rel(gr.cpu().numpy(), g[f'grad{i}']):.2e}", flush=True)


if __name__ == "__main__":
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    encoder()
    detector()
