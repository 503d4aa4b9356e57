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
