"""CPU: samplers against index lists produced by the reference's own sampler classes (tests/golden/
make_sampler_golden.py), collate, and the loader feeding the model kwargs."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))


def test_samplers_match_reference_classes():
    from make_sampler_golden import CASES, DS
    from vidar_amd.data.loader import DistributedGroupSampler, DistributedSampler
    gold = json.loads((GOLD / "sampler.json").read_text())
    for n, spg, world, seed in CASES:
        for epoch in (0, 3):
            seen = []
            for rank in range(world):
                s = DistributedGroupSampler(DS(n), spg, world, rank, seed)
                s.set_epoch(epoch)
                got = list(s)
                assert got == gold[f"train/{n}/{spg}/{world}/{seed}/{epoch}/{rank}"] and len(got) == len(s)
                seen += got
            assert set(seen) == set(range(n))                  # every sample lands on some rank
        for rank in range(world):
            assert list(DistributedSampler(DS(n), world, rank)) == gold[f"test/{n}/{world}/{rank}"]


def test_loader_batches_are_forward_train_kwargs(tmp_path):
    from test_reader_cpu import _mini_nuscenes
    from vidar_amd.data.loader import build_dataloader
    from vidar_amd.data.reader import ViDARSequenceDataset
    ds = ViDARSequenceDataset(_mini_nuscenes(tmp_path), queue_length=2, future_length=1)
    seen = []
    for rank in range(2):
        dl = build_dataloader(ds, samples_per_gpu=1, workers_per_gpu=0, num_replicas=2, rank=rank, seed=0)
        for batch in dl:
            assert batch["img"].shape[:3] == (1, 3, 2) and len(batch["img_metas"]) == 1 and len(batch["gt_points"]) == 1
            seen.append(batch["img_metas"][0][2]["sample_idx"])
    assert len(seen) == 8 and len(set(seen)) == 7              # 7 usable samples, padded to 2 x 4


def test_worker_seeds_differ_across_ranks_and_workers_and_collate_pads():
    """worker k of every rank must not share a numpy stream (torch seeds the ranks identically): the reference's
    num_workers * rank + worker_id + seed; samples whose augmentation drew different sizes are zero-padded."""
    import torch
    from vidar_amd.data.loader import collate, worker_seed
    seeds = {worker_seed(4, r, w, 7) for r in range(8) for w in range(4)}
    assert len(seeds) == 32 and worker_seed(4, 0, 0, 7) == 7 and worker_seed(4, 2, 3, 7) == 18
    a = dict(img=torch.ones(5, 6, 3, 10, 12), img_metas={}, gt_points=torch.zeros(3, 5))
    b = dict(img=torch.ones(5, 6, 3, 8, 16), img_metas={}, gt_points=torch.zeros(4, 5))
    out = collate([a, None, b])
    assert out["img"].shape == (2, 5, 6, 3, 10, 16)
    assert float(out["img"][0, ..., :, 12:].abs().sum()) == 0 and float(out["img"][1, ..., 8:, :].abs().sum()) == 0
    assert float(out["img"][0, ..., :10, :12].min()) == 1 and len(out["gt_points"]) == 2
