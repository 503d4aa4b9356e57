"""Thin launcher (the reference: tools/train.py + tools/dist_train.sh around mmcv's runner):

    python tools/train.py vidar_1_8_nusc_1future --iters 100 --work-dir work_dirs/demo
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py \
        /path/to/released/config.py --iters 1000 --cfg-options model.supervise_all_future=False

CONFIG is one of the in-repo names (vidar_amd.configs.VARIANTS) or a path to a released mmcv-style
config file, which loads unchanged.  Data: the synthetic generator (every rank draws its own samples), or with
--ann-file an info pkl read by vidar_amd.data (multi-sweep lidar, image augmentation, DistributedGroupSampler).  DDP over RCCL, AdamW + cosine/warm-up, grad-clip 35, JSON-lines
log, mmcv-layout checkpoints, --resume-from."""
import argparse
import ast
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--work-dir", default="work_dirs/run")
    ap.add_argument("--resume-from")
    ap.add_argument("--no-backbone", action="store_true")
    ap.add_argument("--rays-per-frame", type=int, default=30000)
    ap.add_argument("--samples", type=int, default=4, help="distinct synthetic samples cycled per rank")
    ap.add_argument("--cfg-options", nargs="*", default=[], help="dotted overrides a.b=c")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ann-file", help="nuScenes / OpenScene info pkl (data/nuscenes/nuscenes_infos_temporal_train.pkl)")
    ap.add_argument("--data-root", default="")
    ap.add_argument("--workers", type=int, default=4)
    args = ap.parse_args()

    from vidar_amd import checkpoint as C
    from vidar_amd import train as T
    from vidar_amd.configs import VARIANTS, get_config
    from vidar_amd.plugin.config import Config
    from vidar_amd.synthetic import fpn_features, make_sample

    rank, local, world = T.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    file_cfg = None
    if args.config in VARIANTS:
        meta = get_config(args.config, with_backbone=not args.no_backbone)
        model_cfg, opt_cfg, clip = meta["model"], meta["optimizer"], meta["grad_clip"]
    else:
        cfg = file_cfg = Config.fromfile(args.config)
        cfg.merge_from_dict({k: ast.literal_eval(v) if v[:1] in "-0123456789[({TFN'\"" else v
                             for k, v in (o.split("=", 1) for o in args.cfg_options)})
        model_cfg = dict(cfg.model)
        name = Path(args.config).stem
        meta = get_config(name if name in VARIANTS else "vidar_1_8_nusc_1future")
        opt_cfg = dict(lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay)
        clip = cfg.optimizer_config.grad_clip.max_norm
        if args.no_backbone:
            model_cfg.pop("img_backbone", None); model_cfg.pop("img_neck", None)
    from vidar_amd import gemm_tuning
    gemm_tuning.enable(rank=rank)               # tuned library-GEMM solutions (vidar_amd/gemm_tuning.py)
    torch.manual_seed(args.seed); np.random.seed(args.seed + rank)
    model = T.build_model(model_cfg).to(dev).train()
    ddp = T.wrap_ddp(model, local)
    opt = T.build_optimizer(model, **opt_cfg)
    sched = T.CosineWithWarmup(opt, args.iters)
    it0 = 0
    if args.resume_from:                 # continue the SAME run: global iteration, schedule position
        _, it0 = C.resume(model, opt, args.resume_from, map_location=dev)
        sched.it = it0
        if it0 >= args.iters:
            raise SystemExit(f"checkpoint is at iteration {it0}, --iters {args.iters} leaves nothing to do")

    def sample(i):
        metas, gt = make_sample(1000 * rank + i, queue_length=meta["queue_length"],
                                future_frames=meta["future_frames"], rays_per_frame=args.rays_per_frame,
                                num_cams=meta["num_cams"], img_hw=meta["img_hw"])
        b = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt).to(dev)])
        if args.no_backbone:
            b["img_feats"] = fpn_features(i, 5, num_cams=meta["num_cams"], shapes=meta["fpn_shapes"], device=dev)
        else:
            g = torch.Generator().manual_seed(1000 * rank + i)
            b["img"] = torch.randn(1, 5, meta["num_cams"], 3, *meta["img_hw"], generator=g).to(dev)
        return b

    work = Path(args.work_dir)
    if rank == 0:
        work.mkdir(parents=True, exist_ok=True)
    if args.ann_file:                      # real data: reader -> rank-sharded sampler -> collate -> device
        from vidar_amd.configs import dataset_kwargs
        from vidar_amd.data import ViDARSequenceDataset
        from vidar_amd.data.loader import build_dataloader
        # the recipe's own temporal augmentation / subset stride / GT voxel size (not the reader's defaults)
        ds = ViDARSequenceDataset(args.ann_file, data_root=args.data_root, augment=True,
                                  **dataset_kwargs(meta, test_mode=False, file_cfg=file_cfg))
        spg = file_cfg.data.samples_per_gpu if file_cfg is not None else meta["data"]["samples_per_gpu"]
        loader = build_dataloader(ds, spg, args.workers, world, rank, args.seed)

        class _Epochs:                      # fit() re-iterates `batches` until --iters: one pass = one epoch
            epoch = 0

            def __iter__(self):
                loader.sampler.set_epoch(self.epoch); self.epoch += 1
                for b in loader:
                    yield dict(img=b["img"].to(dev, non_blocking=True), img_metas=b["img_metas"],
                               gt_points=[g.to(dev, non_blocking=True) for g in b["gt_points"]])
        batches = _Epochs()
    else:
        batches = [sample(i) for i in range(args.samples)]
    n = T.fit(ddp, opt, batches, args.iters, scheduler=sched, max_norm=clip,
              log_every=10, log_path=work / "log.jsonl", ckpt_path=work / "latest.pth",
              ckpt_every=max(1, args.iters // 2), rank=rank, start_iter=it0)
    if rank == 0:
        print(f"finished {n} iterations; log: {work / 'log.jsonl'}; checkpoint: {work / 'latest.pth'}")


if __name__ == "__main__":
    main()
