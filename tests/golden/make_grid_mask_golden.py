"""Golden masks from the reference's own GridMask (projects/mmdet3d_plugin/models/utils/grid_mask.py),
imported in this container (mmcv.runner decorators stubbed, the hard-coded `.cuda()` made a no-op).
    python tests/golden/make_grid_mask_golden.py   ->  tests/golden/grid_mask.npz"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
CASES = [  # (seed, n, c, h, w, cfg)
    (0, 2, 3, 32, 48, dict(use_h=True, use_w=True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)),
    (1, 1, 3, 61, 40, dict(use_h=True, use_w=True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)),
    (2, 2, 2, 29, 50, dict(use_h=True, use_w=False, rotate=1, offset=False, ratio=0.3, mode=0, prob=1.0)),
    (3, 1, 4, 24, 24, dict(use_h=False, use_w=True, rotate=1, offset=True, ratio=0.6, mode=0, prob=1.0)),
    (4, 1, 1, 40, 40, dict(use_h=True, use_w=True, rotate=30, offset=False, ratio=0.5, mode=1, prob=1.0)),
    (5, 3, 3, 32, 48, dict(use_h=True, use_w=True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)),
    (7, 3, 3, 32, 48, dict(use_h=True, use_w=True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)),
]


def main():
    import ref_import as R
    R.install_stubs()
    torch.Tensor.cuda = lambda self, *a, **k: self
    G = R.load_file("ref_grid_mask", R.PLUGIN / "models/utils/grid_mask.py").GridMask
    out = {}
    for i, (seed, n, c, h, w, cfg) in enumerate(CASES):
        np.random.seed(seed)
        x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(seed))
        m = G(**cfg).train()
        out[f"y{i}"] = m(x.clone()).numpy()
        out[f"next_rand{i}"] = np.random.rand()      # generator state after the call: same number of draws
    np.savez_compressed(HERE / "grid_mask.npz", **out)


if __name__ == "__main__":
    main()
