"""CPU: vidar_amd's device-side GridMask against the output of the reference's own GridMask class
(tests/golden/make_grid_mask_golden.py; models/utils/grid_mask.py:69-124) for seeded numpy draws --
same masks, same number of random draws consumed, identity in eval mode; and the detector applies it
where the reference does (detectors/vidar.py:139-140)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))


@pytest.mark.parametrize("i", range(7))
def test_matches_reference_class(i):
    from make_grid_mask_golden import CASES
    from vidar_amd.plugin.utils.grid_mask import GridMask
    gold = np.load(GOLD / "grid_mask.npz")
    seed, n, c, h, w, cfg = CASES[i]
    np.random.seed(seed)
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(seed))
    y = GridMask(**cfg).train()(x.clone())
    np.testing.assert_array_equal(y.numpy(), gold[f"y{i}"])
    assert np.random.rand() == float(gold[f"next_rand{i}"])


def test_eval_mode_is_identity_and_consumes_one_draw():
    from vidar_amd.plugin.utils.grid_mask import GridMask
    x = torch.randn(2, 3, 8, 8)
    np.random.seed(0)
    a = np.random.rand(); b = np.random.rand()
    np.random.seed(0)
    assert GridMask(True, True, prob=1.0).eval()(x) is x
    assert np.random.rand() == b and a != b


def test_detector_masks_the_images_in_training_only():
    from vidar_amd.configs import get_config
    from vidar_amd import train as T
    cfg = get_config("vidar_1_8_nusc_1future", bev_h=24, bev_w=24, with_backbone=True)
    assert cfg["model"]["use_grid_mask"] is True          # every released config enables it
    model = T.build_model(cfg)
    seen = []
    model.img_backbone.forward = lambda img: (seen.append(img.clone()), [])[1]
    model.img_neck = None
    img = torch.ones(1, 2, 3, 32, 48)
    np.random.seed(3)                                     # rand() = 0.55 < prob 0.7 -> masked
    model.train(); model.extract_feat(img.clone())
    model.eval(); model.extract_feat(img.clone())
    assert 0 < float(seen[0].mean()) < 1 and float(seen[1].mean()) == 1.0
    assert torch.equal(seen[0][0], seen[0][1])            # one mask for all images of the call, like the reference
