"""CPU: evaluation loop pieces -- metric summary vs the reference's own `evaluate` (golden made by
tests/golden/make_evaluate_golden.py), sharding + result collection over a 2-rank gloo group, and
the submission writer of `ViDAR.forward_test` (vidar.py:503-519)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_ddp_cpu import _free_port
from test_plugin_cpu import _small_batch

GOLD = Path(__file__).parent / "golden"


def test_summarize_matches_reference_evaluate():
    from vidar_amd.evaluate import format_summary, summarize
    g = json.loads((GOLD / "evaluate_summary.json").read_text())
    got = summarize(g["results"])
    assert got.keys() == g["expected"].keys()
    for fk, fr in g["expected"].items():
        assert got[fk].keys() == fr.keys()
        for k, v in fr.items():
            assert got[fk][k] == v, (fk, k)            # same operations in the same order: exact
    assert g["results"][0]["frame.0"]["count"] in (1, 2)  # inputs are not mutated
    assert "==== frame.0 results: ====" in format_summary(got)
    assert summarize([]) == {}


@pytest.mark.parametrize("size,world", [(7, 2), (8, 2), (1, 2), (0, 2), (5, 3), (4, 1)])
def test_shard_indices_cover_the_dataset_in_rank_strided_order(size, world):
    from vidar_amd.evaluate import shard_indices
    parts = [shard_indices(size, r, world) for r in range(world)]
    assert len({len(p) for p in parts}) == 1                       # padded to equal length
    inter = [p[i] for i in range(len(parts[0])) for p in parts]    # what collect_results rebuilds
    assert inter[:size] == list(range(size))


def _collect_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from vidar_amd.evaluate import multi_gpu_test, summarize
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Fake(torch.nn.Module):                 # stands in for ViDAR.forward_test
        def forward(self, return_loss=True, idx=None):
            assert not return_loss and not self.training
            return [{"frame.0": dict(count=1, chamfer_distance=float(idx), l1_error=2.0 * idx,
                                     absrel_error=0.5)}]
    res = multi_gpu_test(Fake(), lambda i: dict(idx=i), size=7)
    if rank == 0:
        out["order"] = [r["frame.0"]["chamfer_distance"] for r in res]
        out["summary"] = summarize(res)["frame.0"]
    else:
        out["other"] = res
    dist.barrier()
    dist.destroy_process_group()


def test_multi_gpu_test_two_ranks_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_collect_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out["order"] == [float(i) for i in range(7)]            # dataset order, padding dropped
    assert out["other"] is None
    s = out["summary"]
    assert s["count"] == 7 and s["chamfer_distance"] == 3.0 and s["l1_error"] == 6.0 and s["absrel_error"] == 0.5


def test_forward_test_writes_submission_files(tmp_path):
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import fpn_features, make_sample
    torch.manual_seed(0); np.random.seed(0)
    cfg = get_config("vidar_1_8_nusc_3future", bev_h=24, bev_w=24)
    cfg["model"]["test_future_frame_num"] = 2
    cfg["model"]["_submission"] = True
    cfg["model"]["_submission_path"] = str(tmp_path / "sub" / "model")
    n_test = cfg["model"]["test_future_frame_num"]
    metas, gt = make_sample(5, rays_per_frame=150, future_frames=n_test)
    batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt)],
                 img_feats=fpn_features(0, 5, shapes=[(15, 25), (8, 13), (4, 7), (2, 4)]))
    model = T.build_model(cfg)
    with cpu_ops.patched(), torch.no_grad():
        res = model(return_loss=False, **batch)[0]
    assert set(res) == {f"frame.{i}" for i in range(n_test + 1)}
    files = sorted(p.name for p in (tmp_path / "sub" / "model").iterdir())
    assert files == [f"synthetic000005_{f}.txt" for f in range(1, n_test + 1)]   # frame 0 is not submitted
    lines = (tmp_path / "sub" / "model" / files[0]).read_text().splitlines()
    assert len(lines) == 150                                     # one depth per GT ray of that frame
    d = np.array([float(x) for x in lines])
    assert np.all(np.isfinite(d)) and np.all(d >= 0)
    assert all(len(x.split(".")[1]) == 6 for x in lines)         # '%f'
