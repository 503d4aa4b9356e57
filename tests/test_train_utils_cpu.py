"""CPU: checkpoint layout / key compatibility, lr schedule of the released configs, thin fit loop."""
import json

import numpy as np
import torch

from vidar_amd import checkpoint as C
from vidar_amd import train as T
from test_plugin_cpu import _small_batch


def test_checkpoint_roundtrip_and_ddp_prefix(tmp_path):
    cfg, _ = _small_batch("vidar_1_8_nusc_1future")
    torch.manual_seed(0)
    a = T.build_model(cfg); opt = T.build_optimizer(a)
    path = C.save_checkpoint(a, tmp_path / "ckpt.pth", opt, meta=dict(epoch=3, iter=77))
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"meta", "state_dict", "optimizer"} and ck["meta"]["epoch"] == 3
    # a checkpoint written from a DDP-wrapped model carries 'module.' prefixes
    ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}
    torch.save(ck, tmp_path / "ddp.pth")
    torch.manual_seed(1)
    b = T.build_model(cfg)
    _, missing, unexpected = C.load_checkpoint(b, tmp_path / "ddp.pth")
    assert not missing and not unexpected
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    assert C.resume(b, T.build_optimizer(b), path) == (3, 77)


def test_partial_checkpoint_reports_mismatches(tmp_path):
    cfg, _ = _small_batch("vidar_1_8_nusc_1future")
    m = T.build_model(cfg)
    sd = {k: v for k, v in m.state_dict().items() if "latent_render" not in k}
    sd["pts_bbox_head.cls_branches.0.weight"] = torch.zeros(1)      # detection branch ViDAR deletes
    torch.save(dict(state_dict=sd), tmp_path / "p.pth")
    _, missing, unexpected = C.load_checkpoint(m, tmp_path / "p.pth")
    assert all("latent_render" in k for k in missing) and len(missing) == 6
    assert unexpected == ["pts_bbox_head.cls_branches.0.weight"]


def test_cosine_with_warmup_matches_config_values():
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    s = T.CosineWithWarmup(opt, total_iters=10000, warmup_iters=500, warmup_ratio=1 / 3, min_lr_ratio=1e-3)
    lrs = []
    for _ in range(10001):
        s.step(); lrs.append(opt.param_groups[0]["lr"])
    assert abs(lrs[0] - 2e-4 / 3) < 1e-9                       # warm-up starts at ratio 1/3
    assert lrs[499] < lrs[500] and max(lrs) <= 2e-4 + 1e-12
    assert abs(lrs[-1] - 2e-7) < 1e-10                         # floor = 1e-3 * lr
    assert all(a >= b - 1e-15 for a, b in zip(lrs[500:-1], lrs[501:]))


def test_fit_loop_logs_and_checkpoints(tmp_path):
    from oracle import cpu_ops
    torch.manual_seed(0); np.random.seed(0)
    cfg, batch = _small_batch("vidar_1_8_nusc_1future")
    model = T.build_model(cfg).train(); opt = T.build_optimizer(model)
    with cpu_ops.patched():
        n = T.fit(model, opt, [batch], iters=2, scheduler=T.CosineWithWarmup(opt, 2, warmup_iters=1),
                  log_every=1, log_path=tmp_path / "log.jsonl", ckpt_path=tmp_path / "c.pth", ckpt_every=2)
    assert n == 2
    lines = [json.loads(l) for l in open(tmp_path / "log.jsonl")]
    assert [l["iter"] for l in lines] == [1, 2] and "frame.3.regularization.loss.loss" in lines[0]
    assert (tmp_path / "c.pth").exists()
    # resuming continues the SAME run: global iteration in the log / checkpoint, stops at `iters`
    from vidar_amd import checkpoint as C
    model2 = T.build_model(cfg).train(); opt2 = T.build_optimizer(model2)
    _, it0 = C.resume(model2, opt2, tmp_path / "c.pth")
    assert it0 == 2
    with cpu_ops.patched():
        n = T.fit(model2, opt2, [batch], iters=3, log_every=1, log_path=tmp_path / "log.jsonl",
                  ckpt_path=tmp_path / "c.pth", ckpt_every=1, start_iter=it0)
    assert n == 3
    assert [json.loads(l)["iter"] for l in open(tmp_path / "log.jsonl")] == [1, 2, 3]
    assert C.resume(T.build_model(cfg), None, tmp_path / "c.pth")[1] == 3


def test_state_dict_layout_does_not_depend_on_init_weights():
    """released checkpoints carry no pts_bbox_head.transformer.reference_points.* (the reference deletes the
    Linear inside init_weights, vidar_bevformer_head.py:20-23): same layout with or without that call."""
    from vidar_amd import plugin
    cfg, _ = _small_batch("vidar_1_8_nusc_1future")
    a = plugin.build_detector(dict(cfg["model"]))
    keys = set(a.state_dict())
    assert not [k for k in keys if "reference_points" in k]
    a.init_weights()
    assert set(a.state_dict()) == keys


def test_overfitting_one_sample_reduces_the_loss():
    """8 AdamW steps on one synthetic sample (ops on the CPU oracle): the summed loss must fall
    steadily -- every custom backward on the path points downhill end to end."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).parent))
    import numpy as np
    from oracle import cpu_ops
    from test_plugin_cpu import _small_batch
    from vidar_amd import train as T
    torch.manual_seed(0); np.random.seed(0)
    cfg, batch = _small_batch("vidar_1_8_nusc_1future")
    model = T.build_model(cfg).train()
    for m in model.modules():
        if hasattr(m, "random_drop_prev_rate"):
            m.random_drop_prev_rate = 0.0
    opt = T.build_optimizer(model, lr=1e-3)
    losses = []
    with cpu_ops.patched():
        for _ in range(8):
            loss, _ = T.train_step(model, opt, batch)
            losses.append(float(loss))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.9 * losses[0], losses
    assert sum(b < a for a, b in zip(losses, losses[1:])) >= 6, losses
