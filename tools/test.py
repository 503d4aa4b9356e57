"""Thin evaluation launcher (the reference: tools/test.py + tools/dist_test.sh around
`custom_multi_gpu_test`, projects/mmdet3d_plugin/bevformer/apis/test.py:45-114):

    python tools/test.py vidar_1_8_nusc_3future --samples 8 [--checkpoint work_dirs/demo/latest.pth]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/test.py \
        vidar_1_8_nusc_3future --samples 64 --submission submission/model

Every rank evaluates samples rank, rank+W, ... (synthetic generator, or with --ann-file the validation info pkl
read by vidar_amd.data in test mode: full history required, key-frame cloud only, no augmentation), the
per-sample `frame.k` dicts are gathered over the process group (RCCL) and rank 0 prints the
reference's summary (chamfer distance, L1 and AbsRel ray errors per future frame)."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--checkpoint")
    ap.add_argument("--samples", type=int, default=4, help="size of the synthetic evaluation set")
    ap.add_argument("--no-backbone", action="store_true")
    ap.add_argument("--rays-per-frame", type=int, default=30000)
    ap.add_argument("--bev", type=int, nargs=2, help="evaluate at a reduced BEV (plumbing runs)")
    ap.add_argument("--submission", help="directory for the per-sample depth files (vidar.py:503-519)")
    ap.add_argument("--out", help="write the summary as JSON")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ann-file", help="validation info pkl (data/nuscenes/nuscenes_infos_temporal_val.pkl)")
    ap.add_argument("--data-root", default="")
    ap.add_argument("--eval-stride", type=int, default=None,
                    help="evaluate every N-th usable val index (quick subset); default: the whole split, as the "
                         "reference's data.test does")
    args = ap.parse_args(argv)

    from vidar_amd import checkpoint as C
    from vidar_amd import evaluate as E
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import fpn_features, make_sample

    rank, local, world = T.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("tools/test.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    kw = dict(bev_h=args.bev[0], bev_w=args.bev[1]) if args.bev else {}
    meta = get_config(args.config, with_backbone=not args.no_backbone, **kw)
    if args.submission:
        meta["model"]["_submission"] = True
        meta["model"]["_submission_path"] = args.submission
    torch.manual_seed(args.seed); np.random.seed(args.seed)
    model = T.build_model(meta["model"]).to(dev)
    if args.checkpoint:
        C.load_checkpoint(model, args.checkpoint, map_location=dev)
    n_future = meta["model"]["test_future_frame_num"]

    def batch(i):
        metas, gt = make_sample(50000 + i, queue_length=meta["queue_length"], future_frames=n_future,
                                rays_per_frame=args.rays_per_frame, num_cams=meta["num_cams"],
                                img_hw=meta["img_hw"])
        b = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt).to(dev)])
        T_img = meta["queue_length"] + 1
        if args.no_backbone:
            b["img_feats"] = fpn_features(i, T_img, num_cams=meta["num_cams"], shapes=meta["fpn_shapes"],
                                          device=dev)
        else:
            g = torch.Generator().manual_seed(50000 + i)
            b["img"] = torch.randn(1, T_img, meta["num_cams"], 3, *meta["img_hw"], generator=g).to(dev)
        return b

    n_samples = args.samples
    if args.ann_file:
        from vidar_amd.configs import dataset_kwargs
        from vidar_amd.data import ViDARSequenceDataset
        # the whole val split, like the reference's data.test (no load_frame_interval there); --eval-stride N opts into
        # a 1/N subset for a quick look
        kw = dataset_kwargs(meta, test_mode=True, test_stride=args.eval_stride)
        kw["future_length"] = n_future
        ds = ViDARSequenceDataset(args.ann_file, data_root=args.data_root, **kw)
        n_samples = len(ds) if args.samples <= 0 else min(args.samples, len(ds))

        def batch(i):                                                  # noqa: F811  (real data replaces the generator)
            s_ = ds[i]
            return dict(img=s_["img"][None].to(dev), img_metas=[s_["img_metas"]], gt_points=[s_["gt_points"].to(dev)])
    results = E.multi_gpu_test(model, batch, n_samples)
    if rank == 0:
        summary = E.summarize(results)
        print(E.format_summary(summary), flush=True)
        if args.out:
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.out).write_text(json.dumps(summary, indent=1))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
