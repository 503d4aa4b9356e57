"""Evaluation loop pieces around `ViDAR.forward_test`: per-rank test loop, result collection over
the process group and the metric summary.

Mirrors (behaviour, not code):
  * `NuScenesViDARDatasetTemplate.evaluate` (projects/mmdet3d_plugin/datasets/
    nuscenes_vidar_dataset_template.py:147-196): sum the per-sample `frame.k` dicts, divide every
    metric by the frame's `count`.
  * `custom_multi_gpu_test` + `collect_results_cpu` (projects/mmdet3d_plugin/bevformer/apis/
    test.py:45-162): rank r evaluates samples r, r+W, r+2W, ...; the parts are interleaved back into
    dataset order and truncated to the dataset size (the sampler pads the last round).  The
    reference exchanges pickles through a shared tmpdir; here it is one `all_gather_object` over
    the process group (RCCL on GPU ranks, gloo on CPU).
"""
from __future__ import annotations

import copy
from typing import Callable, Iterable, List, Sequence

import torch
import torch.distributed as dist


def summarize(results: Sequence[dict]) -> dict:
    """list of {'frame.k': {count, chamfer_distance, l1_error, absrel_error}} -> per-frame means."""
    if len(results) == 0:
        return {}
    total = copy.deepcopy(results[0])
    for res in results[1:]:
        for frame_k, frame_res in res.items():
            for k, v in frame_res.items():
                total[frame_k][k] += v
    for frame_k, frame_res in total.items():
        count = frame_res["count"]
        for k in frame_res:
            if k != "count":
                frame_res[k] = frame_res[k] / count
    return total


def format_summary(summary: dict) -> str:
    lines = []
    for frame_k, frame_res in summary.items():
        lines.append(f"==== {frame_k} results: ====")
        lines += [f"{k}: {v}" for k, v in frame_res.items()]
    return "\n".join(lines)


def collect_results(result_part: List, size: int) -> List | None:
    """Gather every rank's result list; rank 0 returns them in dataset order, other ranks None."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(result_part)[:size]
    world = dist.get_world_size()
    parts: List = [None] * world
    dist.all_gather_object(parts, list(result_part))
    if dist.get_rank() != 0:
        return None
    ordered = []
    for i in range(max(len(p) for p in parts)):
        for p in parts:
            if i < len(p):
                ordered.append(p[i])
    return ordered[:size]


def shard_indices(size: int, rank: int, world: int) -> List[int]:
    """Test-time DistributedSampler(shuffle=False): pad to a multiple of `world` by wrapping around,
    rank r takes r, r+W, ..."""
    if size == 0:
        return []
    per = (size + world - 1) // world
    padded = list(range(size)) + [i % size for i in range(per * world - size)]
    return padded[rank::world]


@torch.no_grad()
def multi_gpu_test(model, get_batch: Callable[[int], dict], size: int) -> List | None:
    """Every rank runs forward_test on its shard; rank 0 gets all results in dataset order."""
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    model.eval()
    part = []
    for i in shard_indices(size, rank, world):
        out = model(return_loss=False, **get_batch(i))
        part.extend(out)
    return collect_results(part, size)
