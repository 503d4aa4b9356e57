"""GPU: host<->device synchronisation points of one training step, counted with
torch.cuda.set_sync_debug_mode("warn").  The reference synchronises per camera per layer (nonzero() at
spatial_cross_attention.py:138, boolean indexing at vidar_head_base.py:441, :464-467, :636-644); DDP scaling
needs the step to run ahead of the GPU, so the budget here is the ONE planned read of the visible-query list
lengths (encoder.plan_frames); one more is tolerated."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))


@pytest.mark.parametrize("name", ["vidar_1_8_nusc_1future", "vidar_1_8_nusc_3future"])
def test_at_most_two_host_syncs_per_training_step(name):
    from sync_count import count_syncs
    from test_plugin_cpu import _small_batch
    from vidar_amd import train as T
    torch.manual_seed(0); np.random.seed(0)
    cfg, batch = _small_batch(name)
    model = T.build_model(cfg).cuda().train()
    opt = T.build_optimizer(model)
    batch = dict(img_metas=batch["img_metas"], gt_points=[g.cuda() for g in batch["gt_points"]],
                 img_feats=[f.cuda() for f in batch["img_feats"]])
    for _ in range(2):
        T.train_step(model, opt, batch)
    n, where = count_syncs(lambda: T.train_step(model, opt, batch))
    assert n <= 2, dict(where)
