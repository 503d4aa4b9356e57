"""Count host<->device synchronisation points of ONE training step (torch.cuda.set_sync_debug_mode("warn")):
prints every source line that synchronises and how often.  Needs a GPU.
    python tools/sync_count.py [--config vidar_1_8_nusc_3future] [--no-backbone]"""
import argparse
import collections
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def count_syncs(fn):
    """-> (n, Counter{'file:line': n}) of synchronising calls made by fn()"""
    where = collections.Counter()
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        fn()
        torch.cuda.set_sync_debug_mode("default")
    n = 0
    for w in rec:
        msg = str(w.message)
        if "prototype feature" in msg:           # set_sync_debug_mode's own notice, not a synchronisation
            continue
        if "synchroniz" in msg:
            n += 1
            where[f"{w.filename}:{w.lineno}"] += 1
    return n, where


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vidar_1_8_nusc_1future")
    ap.add_argument("--no-backbone", action="store_true")
    ap.add_argument("--bev", type=int, default=200)
    args = ap.parse_args()
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import fpn_features, make_sample
    dev = torch.device("cuda", 0)
    cfg = get_config(args.config, bev_h=args.bev, bev_w=args.bev, with_backbone=not args.no_backbone)
    torch.manual_seed(0); np.random.seed(0)
    model = T.build_model(cfg).to(dev).train()
    opt = T.build_optimizer(model)
    metas, gt = make_sample(0, future_frames=cfg["future_frames"], num_cams=cfg["num_cams"], img_hw=cfg["img_hw"])
    batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt).to(dev)])
    if args.no_backbone:
        batch["img_feats"] = fpn_features(0, 5, num_cams=cfg["num_cams"], shapes=cfg["fpn_shapes"], device=dev)
    else:
        batch["img"] = torch.randn(1, 5, cfg["num_cams"], 3, *cfg["img_hw"]).to(dev)
    for _ in range(2):
        T.train_step(model, opt, batch)
    n, where = count_syncs(lambda: T.train_step(model, opt, batch))
    print(f"{n} synchronising calls in one training step of {args.config}")
    for k, v in where.most_common():
        print(f"  {v:4d}  {k}")


if __name__ == "__main__":
    main()
