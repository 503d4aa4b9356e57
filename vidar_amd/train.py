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
    # torch's single-kernel AdamW instead of the foreach implementation's ~12 elementwise launches per step (same fp32
    # update, different instruction order); VIDAR_FUSED_ADAMW=0 restores the foreach form
    fused = os.environ.get("VIDAR_FUSED_ADAMW", "1") != "0" and all(p.is_cuda for g in groups for p in g["params"])
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, **({"fused": True} if fused else {}))


def init_distributed():
    """torchrun-style env -> (rank, local_rank, world).  Backend 'nccl' == RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VIDAR_FORCE_DDP=1: also form a 1-rank group (exercises RCCL + the DDP reducer on a 1-GPU box)
    forced = os.environ.get("VIDAR_FORCE_DDP") == "1" and "RANK" in os.environ
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatAllReduce(torch.nn.Module):
    """Data parallelism with ONE gradient all-reduce per step on a flat buffer, after backward.

    What torch's DistributedDataParallel costs on this step was measured on one rank of a real RCCL group
    (tools/ddp_overhead.py, profiles/r05_ddp_overhead_1rank.json): +5.9 ms per step, of which the reducer's autograd
    hooks are 0.6 ms and the collective itself 0.09 ms -- the rest is the per-parameter bucket traffic of a reducer that
    overlaps the all-reduce with backward (334 gradients copied into bucket views one launch at a time inside a
    backward pass that is GPU-bound), i.e. the overlap machinery costs more than the 250 MB all-reduce it hides (~2 ms
    over 8 x xGMI, < 1 % of a 350 ms step).  So: no buckets -- after backward the gradients are gathered into
    one flat buffer (one batched copy), pre-divided by the world size, all-reduced (RCCL over xGMI) and handed back to
    the parameters as views of that buffer.  Same result as DDP's averaged gradients (sum of g / world in rank order
    is what both compute).  `VIDAR_DDP=torch` selects torch's DistributedDataParallel instead.

    mode "flat2" (`VIDAR_DDP=flat2`): TWO flat buffers.  The image backbone + neck hold 2/3 of the gradient bytes and
    their backward finishes in the MIDDLE of the step's backward pass: forward_train computes the history BEV (whose
    encoder pass is back-propagated, detectors/vidar.py:273-287) BEFORE the current frame's backbone, so autograd
    runs head -> decoder -> encoder -> backbone -> history encoder.  A post-accumulate hook per backbone / neck parameter
    counts the gradients in; when the last one has arrived their flat buffer is all-reduced ASYNCHRONOUSLY (RCCL runs the
    collective on its own stream) under the history encoder's backward, and only the remaining third is exchanged after
    backward.  Same sums in the same rank order -> the same update as "flat" (tests/test_ddp_cpu.py).  Which of flat /
    flat2 / torch is fastest at 8 ranks is a measurement bench.py makes itself whenever it runs on more than one rank
    (`ddp_ab` in its line); on one rank the three differ only by their host overhead."""

    def __init__(self, module, mode="flat"):
        super().__init__()
        assert mode in ("flat", "flat2")
        self.module = module
        self.mode = mode
        self.world = dist.get_world_size()
        self.last_bytes = 0
        self.early_bytes = 0
        self.early_was_async = False
        with torch.no_grad():                      # every rank starts from rank 0's weights and buffers (DDP does the same)
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0)
        self._early, self._early_ids, self._pending, self._work = [], set(), 0, None
        if mode == "flat2":
            self._early = [p for n, p in module.named_parameters()
                           if p.requires_grad and n.startswith(("img_backbone.", "img_neck."))]
            self._early_ids = {id(p) for p in self._early}
            for p in self._early:
                p.register_post_accumulate_grad_hook(self._early_grad_arrived)

    def forward(self, *args, **kwargs):
        self._pending = len(self._early) if torch.is_grad_enabled() else 0
        self._work = None
        return self.module(*args, **kwargs)

    def _early_grad_arrived(self, p):
        if self._pending <= 0:
            return
        self._pending -= 1
        if self._pending == 0:
            self._work = self._start(self._early, async_op=True)

    def _start(self, params, async_op):
        """flat buffer of `params`' gradients (+ one "this rank produced a gradient" float per parameter) -> all-reduce"""
        have = [p.grad is not None for p in params]
        ref = next((p.grad for p in params if p.grad is not None), params[0])
        mask = torch.tensor([1.0 if h else 0.0 for h in have], dtype=ref.dtype, device=ref.device)
        flat = torch.cat([(p.grad if h else torch.zeros_like(p)).reshape(-1) for p, h in zip(params, have)] + [mask])
        n_grad = flat.numel() - len(params)
        if self.world > 1:
            flat[:n_grad].div_(self.world)
        work = dist.all_reduce(flat, async_op=async_op)
        return params, have, flat, n_grad, work

    def _finish(self, started):
        params, have, flat, n_grad, work = started
        if work is not None:
            work.wait()
        nobody = set()
        if not all(have):
            total = flat[n_grad:].cpu()
            nobody = {i for i, h in enumerate(have) if not h and float(total[i]) == 0.0}
        off = 0
        for i, p in enumerate(params):
            n = p.numel()
            p.grad = None if i in nobody else flat[off:off + n].view_as(p)
            off += n
        return flat.numel() * flat.element_size()

    def reduce_gradients(self):
        # every rank must contribute the same layout: all trainable parameters in module order, a parameter that got
        # no gradient on this rank contributes zeros.  One extra float per parameter rides along ("this rank produced a
        # gradient"): a parameter NO rank produced a gradient for keeps grad None afterwards -- the optimizer then skips
        # it exactly like the 1-GPU step does (AdamW would otherwise apply weight decay and decay its moments; the
        # reference's DDP, find_unused_parameters=False (apis/mmdet_train.py:71-79), refuses such a step instead).
        # The mask is only read back (one host sync) on a rank that itself missed a gradient -- never on the hot path,
        # which gives every trainable parameter a gradient every step (tests/test_ddp_cpu.py).
        params = [p for p in self.module.parameters() if p.requires_grad]
        if not params:
            return
        early = self._work
        self.early_was_async = early is not None
        if self._early and early is None:
            # a backbone / neck gradient never arrived on this rank (its hook count did not run out): exchange the early
            # buffer now -- every rank issues the same two collectives in the same order either way
            early = self._start(self._early, async_op=False)
        late = self._start([p for p in params if id(p) not in self._early_ids], async_op=False)
        self.early_bytes = self._finish(early) if early is not None else 0
        self.last_bytes = self._finish(late) + self.early_bytes
        self._work, self._pending = None, 0

    def logging_data(self):
        if self.mode == "flat2":
            return {"mode": "two flat all-reduces: backbone + neck asynchronously under the history encoder's backward, the "
                            "rest after backward (vidar_amd.train.FlatAllReduce, flat2)", "buckets": 2,
                    "bucket_bytes": [self.early_bytes, self.last_bytes - self.early_bytes],
                    "early_bucket_overlapped": self.early_was_async, "allreduce_bytes_per_step": self.last_bytes}
        return {"mode": "flat all-reduce after backward (vidar_amd.train.FlatAllReduce)", "buckets": 1,
                "bucket_bytes": [self.last_bytes], "allreduce_bytes_per_step": self.last_bytes}


DDP_MODES = ("flat", "flat2", "torch")


def wrap_ddp(model, local_rank, bucket_cap_mb=100, mode=None):
    """mode: one of DDP_MODES (default: $VIDAR_DDP, else "flat")"""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    if dist.get_world_size() == 1 and os.environ.get("VIDAR_FORCE_DDP") != "1":
        return model
    mode = mode or os.environ.get("VIDAR_DDP", "flat")
    if mode not in DDP_MODES:
        raise ValueError(f"gradient exchange {mode!r}: expected one of {DDP_MODES}")
    if mode != "torch":
        return FlatAllReduce(model, mode)
    kw = dict(broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
              find_unused_parameters=False)
    if torch.cuda.is_available():
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], **kw)
    return torch.nn.parallel.DistributedDataParallel(model, **kw)


def train_step(model, optimizer, batch, max_norm=35.0):
    """forward -> sum of the loss dict -> backward -> gradient all-reduce (flat, or DDP's buckets) -> clip -> AdamW step."""
    losses = model(return_loss=True, **batch)
    total = sum(v for v in losses.values())
    optimizer.zero_grad(set_to_none=True)
    total.backward()
    if isinstance(model, FlatAllReduce):
        model.reduce_gradients()
    params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    torch.nn.utils.clip_grad_norm_(params, max_norm)
    optimizer.step()
    return total.detach(), {k: v.detach() for k, v in losses.items()}


class CosineWithWarmup:
    """lr_config of the released configs (vidar_1_8_nusc_3future.py:387-395): CosineAnnealing with a
    linear warm-up of `warmup_iters` iterations starting at `warmup_ratio` x lr, floor
    `min_lr_ratio` x lr (mmcv formulas: annealing_cos and the 'linear' warm-up factor
    1 - (1 - it/warmup_iters)(1 - warmup_ratio)).  Stepped once per iteration; mmcv's default for this
    policy anneals per EPOCH (by_epoch=True) with the same formula -- pass total_iters = epochs and
    call step() per epoch to reproduce that."""

    def __init__(self, optimizer, total_iters, warmup_iters=500, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3):
        import math
        self._cos = math.cos
        self._pi = math.pi
        self.opt, self.total = optimizer, max(1, total_iters)
        self.warmup_iters, self.warmup_ratio, self.min_lr_ratio = warmup_iters, warmup_ratio, min_lr_ratio
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.it = 0

    def lr_at(self, it, base):
        target = base * self.min_lr_ratio
        lr = target + 0.5 * (base - target) * (1 + self._cos(self._pi * min(it, self.total) / self.total))
        if it < self.warmup_iters:
            k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
            lr = lr * (1 - k)
        return lr

    def step(self):
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = self.lr_at(self.it, b)
        self.it += 1


def fit(model, optimizer, batches, iters, scheduler=None, max_norm=35.0, log_every=50, log_path=None,
        ckpt_path=None, ckpt_every=0, rank=0, start_iter=0, freeze_gemm_tuning_after=20):
    """Thin training loop (the reference delegates this to mmcv's EpochBasedRunner + hooks, which are
    out of scope): `batches` is any iterable of forward_train kwargs; JSON-lines log of the loss
    terms and samples/s; optional periodic checkpoints in the mmcv dictionary layout.  After
    `freeze_gemm_tuning_after` iterations TunableOp stops tuning new GEMM shapes (gemm_tuning.freeze):
    a rank that met a new padded SpatialCrossAttention length later would otherwise tune for seconds
    while the other ranks wait at the gradient all-reduce."""
    import json
    import time
    from .checkpoint import save_checkpoint
    it = start_iter          # global iteration: a resumed run continues counting (and stops at `iters`)
    t0 = time.perf_counter()
    log = open(log_path, "a") if (log_path and rank == 0) else None
    while it < iters:
        for batch in batches:
            if scheduler is not None:
                scheduler.step()
            total, parts = train_step(model, optimizer, batch, max_norm)
            it += 1
            if it - start_iter == freeze_gemm_tuning_after:
                from . import gemm_tuning
                gemm_tuning.freeze()
            if log is not None and (it % log_every == 0 or it == iters):
                rec = dict(iter=it, lr=optimizer.param_groups[0]["lr"], loss=float(total),
                           samples_per_s_per_rank=(it - start_iter) / (time.perf_counter() - t0),
                           **{k: float(v) for k, v in parts.items()})
                log.write(json.dumps(rec) + "\n"); log.flush()
            if ckpt_path and ckpt_every and it % ckpt_every == 0 and rank == 0:
                save_checkpoint(model, ckpt_path, optimizer, meta=dict(iter=it, epoch=0))
            if it >= iters:
                break
    if log is not None:
        log.close()
    return it
