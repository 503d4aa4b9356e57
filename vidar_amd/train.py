"""Training step + data-parallel launcher glue for the ViDAR hot path.

Mirrors the reference's runtime choices (apis/mmdet_train.py:71-79, config :379-395): one process
per GPU, DDP(broadcast_buffers=False) with gradients all-reduced over RCCL ("nccl" backend on
ROCm), AdamW(lr 2e-4, wd 0.01), grad-clip L2 35."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import plugin
from .plugin.config import Config


def build_model(cfg, bev_hw=None):
    if isinstance(cfg, Config):
        model_cfg = dict(cfg.model)
    else:
        model_cfg = dict(cfg["model"]) if "model" in cfg and "type" not in cfg else dict(cfg)
    if bev_hw is not None:
        model_cfg = _resize_bev(model_cfg, *bev_hw)
    model = plugin.build_detector(model_cfg)
    model.init_weights()
    return model


def _resize_bev(cfg, h, w):
    """config 0 of BASELINE.json: the same model at a smaller BEV (plumbing runs)."""
    import copy
    cfg = copy.deepcopy(cfg)

    def walk(d):
        if isinstance(d, dict):
            for k in list(d.keys()):
                if k == "bev_h": d[k] = h
                elif k == "bev_w": d[k] = w
                elif k == "row_num_embed": d[k] = h
                elif k == "col_num_embed": d[k] = w
                elif k == "rotate_center": d[k] = [w // 2, h // 2]
                else: walk(d[k])
        elif isinstance(d, (list, tuple)):
            for v in d: walk(v)
    walk(cfg)
    return cfg


def build_optimizer(model, lr=2e-4, weight_decay=0.01, backbone_lr_mult=0.1):
    bb, rest = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (bb if n.startswith("img_backbone") else rest).append(p)
    groups = [dict(params=rest)]
    if bb:
        groups.append(dict(params=bb, lr=lr * backbone_lr_mult))
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)


def init_distributed():
    """torchrun-style env -> (rank, local_rank, world).  Backend 'nccl' == RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def wrap_ddp(model, local_rank, bucket_cap_mb=100):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    kw = dict(broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
              find_unused_parameters=False)
    if torch.cuda.is_available():
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], **kw)
    return torch.nn.parallel.DistributedDataParallel(model, **kw)


def train_step(model, optimizer, batch, max_norm=35.0):
    """forward -> sum of the loss dict -> backward (DDP all-reduce) -> clip -> AdamW step."""
    losses = model(return_loss=True, **batch)
    total = sum(v for v in losses.values())
    optimizer.zero_grad(set_to_none=True)
    total.backward()
    params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    torch.nn.utils.clip_grad_norm_(params, max_norm)
    optimizer.step()
    return total.detach(), {k: v.detach() for k, v in losses.items()}
