"""CPU, world_size 2, gloo: the data-parallel path (one process per rank, DDP gradient all-reduce,
per-rank samples, max-over-ranks timing) with the ops routed to the CPU oracle."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out, mode, with_backbone=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), VIDAR_DDP=mode)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import cpu_ops
    from vidar_amd import train as T
    from test_plugin_cpu import _small_batch
    torch.set_num_threads(2)
    r, l, w = T.init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.manual_seed(7)                       # same weights everywhere
    np.random.seed(rank)
    cfg, batch = _small_batch("vidar_1_8_nusc_1future", seed=10 + rank)   # different sample per rank
    if with_backbone:
        from test_weights_cpu import _tiny_image_batch
        cfg, batch = _tiny_image_batch()
        batch["img"] = batch["img"] + 0.1 * rank
    model = T.build_model(cfg).train()
    ddp = T.wrap_ddp(model, l)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel if mode == "torch" else T.FlatAllReduce)
    opt = T.build_optimizer(model)
    with cpu_ops.patched():
        loss, _ = T.train_step(ddp, opt, batch)
    if mode.startswith("flat"):
        info = ddp.logging_data()
        assert info["buckets"] == (2 if mode == "flat2" else 1) and all(b > 0 for b in info["bucket_bytes"])
        assert info["allreduce_bytes_per_step"] == sum(info["bucket_bytes"])
        if mode == "flat2":                      # the early buffer really left from inside backward, not from the fallback
            assert info["early_bucket_overlapped"] is with_backbone
    # after the all-reduced step the replicas are still identical although the samples differ
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    out[rank] = (float(loss), float((gathered[0] - gathered[1]).abs().max()), flat[::997].double().sum().item())
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, with_backbone=False):
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out, mode, with_backbone), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    assert out[0][0] != out[1][0], "ranks must see different samples"
    assert out[0][1] == 0.0, "parameters diverged across ranks after the data-parallel step"
    return out[0][2]


def test_ddp_two_ranks_gloo():
    """the default gradient exchange (one flat all-reduce after backward, train.FlatAllReduce) and torch's
    DistributedDataParallel (VIDAR_DDP=torch): replicas stay identical, and both produce the same update"""
    a = _run("flat")
    b = _run("torch")
    assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)


def test_flat2_overlapped_exchange_gives_the_same_update():
    """VIDAR_DDP=flat2 (backbone + neck gradients all-reduced asynchronously as soon as the last of them has arrived,
    the rest after backward) on the image-in step: replicas stay identical and the update equals the one-buffer form"""
    a = _run("flat", with_backbone=True)
    b = _run("flat2", with_backbone=True)
    assert a == b, (a, b)                         # the same sums in the same rank order


def _unused_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vidar_amd import train as T
    T.init_distributed()

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(3, 1)
            self.only_rank0 = torch.nn.Parameter(torch.ones(3))
            self.never = torch.nn.Parameter(torch.ones(2))

        def forward(self, x, use):
            y = self.a(x).sum()
            return y + (self.only_rank0 * x[0]).sum() if use else y
    torch.manual_seed(0)
    m = T.FlatAllReduce(Toy())
    x = torch.full((2, 3), float(rank + 1))
    m(x, use=rank == 0).backward()
    m.reduce_gradients()
    t = m.module
    out[rank] = (t.never.grad is None, t.only_rank0.grad.tolist(), t.a.weight.grad.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_all_reduce_leaves_grad_none_where_no_rank_produced_one():
    """the 1-GPU step skips a parameter without a gradient (train_step's `p.grad is not None` filter); the flat exchange
    must not turn that into a zero gradient (AdamW would decay the weight), and a gradient only SOME ranks produced is
    still averaged over all of them"""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_unused_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (0, 1):
        never_none, only0, aw = out[r]
        assert never_none
        assert only0 == [0.5, 0.5, 0.5]                    # rank 0: x[0] = 1 -> grad 1; rank 1: none -> 0; mean 0.5
        assert aw == [[3.0, 3.0, 3.0]]                     # (2 * 1 + 2 * 2) / 2
