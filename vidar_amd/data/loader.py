"""Rank-sharded samplers + collate: the last stretch between `ViDARSequenceDataset` and `forward_train`.

  DistributedGroupSampler   projects/mmdet3d_plugin/datasets/samplers/group_sampler.py:12-112 (training): per group an
                            epoch-seeded permutation padded to a multiple of samples_per_gpu x world, then a second
                            permutation of the samples_per_gpu-sized chunks, rank r takes the r-th contiguous block.
                            ViDAR's datasets put every sample in group 0 (mmdet3d `_set_group_flag` [3P]).
  DistributedSampler        samplers/distributed_sampler.py:9-41 (testing): no shuffle, contiguous per-rank shards, padded
                            by repetition.
  collate                   the unwrapping of mmcv DataContainers the reference leaves to mmcv's collate [3P]: images
                            stacked, metas / point clouds as per-sample lists -- the kwargs of ViDAR.forward_train."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.utils.data import Sampler


class DistributedGroupSampler(Sampler):
    def __init__(self, dataset, samples_per_gpu=1, num_replicas=1, rank=0, seed=0):
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.num_replicas, self.rank, self.epoch, self.seed = num_replicas, rank, 0, seed if seed is not None else 0
        self.flag = np.asarray(getattr(dataset, "flag", np.zeros(len(dataset), dtype=np.uint8)))
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(math.ceil(s / samples_per_gpu / num_replicas)) * samples_per_gpu
                               for s in self.group_sizes)
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            idx = np.where(self.flag == i)[0]
            idx = idx[torch.randperm(int(size), generator=g).numpy()].tolist()
            extra = int(math.ceil(size / self.samples_per_gpu / self.num_replicas)) * self.samples_per_gpu \
                * self.num_replicas - len(idx)
            tmp = idx.copy()
            for _ in range(extra // size):
                idx.extend(tmp)
            idx.extend(tmp[:extra % size])
            indices.extend(idx)
        assert len(indices) == self.total_size
        spg = self.samples_per_gpu
        indices = [indices[j] for i in torch.randperm(len(indices) // spg, generator=g).tolist()
                   for j in range(i * spg, (i + 1) * spg)]
        off = self.num_samples * self.rank
        return iter(indices[off:off + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class DistributedSampler(Sampler):
    """test-time shards: rank r evaluates indices [r*n, (r+1)*n) of the (repeat-padded) dataset"""

    def __init__(self, dataset, num_replicas=1, rank=0):
        self.n, self.num_replicas, self.rank = len(dataset), num_replicas, rank
        self.num_samples = int(math.ceil(self.n / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        idx = list(range(self.n))
        idx = (idx * math.ceil(self.total_size / max(len(idx), 1)))[:self.total_size]
        per = self.total_size // self.num_replicas
        return iter(idx[self.rank * per:(self.rank + 1) * per])

    def __len__(self):
        return self.num_samples


def collate(samples):
    """list of dataset samples -> dict(img [bs,T,cams,3,H,W], img_metas [bs] of {t: meta}, gt_points [bs]).
    Images of different sizes (CropResizeFlipImage draws a resize per sample) are zero-padded at the bottom /
    right to the largest H and W of the batch, like mmcv's collate does for stacked DataContainers."""
    samples = [s for s in samples if s is not None]
    imgs = [s["img"] for s in samples]
    H = max(i.shape[-2] for i in imgs); W = max(i.shape[-1] for i in imgs)
    if any(i.shape[-2:] != (H, W) for i in imgs):
        imgs = [torch.nn.functional.pad(i, (0, W - i.shape[-1], 0, H - i.shape[-2])) for i in imgs]
    return dict(img=torch.stack(imgs), img_metas=[s["img_metas"] for s in samples],
                gt_points=[s["gt_points"] for s in samples])


def worker_seed(num_workers, rank, worker_id, seed):
    """mmdet's worker_init_fn (apis of the reference's build_dataloader): num_workers * rank + worker_id + seed"""
    return num_workers * rank + worker_id + seed


def _seed_worker(worker_id, num_workers, rank, seed):
    import random
    s = worker_seed(num_workers, rank, worker_id, seed)
    np.random.seed(s)
    random.seed(s)
    torch.manual_seed(s)


def build_dataloader(dataset, samples_per_gpu=1, workers_per_gpu=4, num_replicas=1, rank=0, seed=0, test_mode=False):
    """every worker of every rank gets its own numpy / random stream (photometric, flip, frame interval draws):
    torch alone would hand worker k of EVERY rank the same seed, since the ranks seed torch identically"""
    from functools import partial
    sampler = DistributedSampler(dataset, num_replicas, rank) if test_mode else \
        DistributedGroupSampler(dataset, samples_per_gpu, num_replicas, rank, seed)
    init = partial(_seed_worker, num_workers=workers_per_gpu, rank=rank, seed=seed)
    return torch.utils.data.DataLoader(dataset, batch_size=samples_per_gpu, sampler=sampler, num_workers=workers_per_gpu,
                                       collate_fn=collate, pin_memory=True, drop_last=False,
                                       worker_init_fn=init if workers_per_gpu > 0 else None)
