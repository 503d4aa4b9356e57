"""Golden for vidar_amd.data.augment.CropResizeFlipImage from the reference's own class
(projects/mmdet3d_plugin/datasets/pipelines/augmentation.py:10-203) executed here (mmcv / mmdet registry stubbed):
seeded draws, two consecutive frames sharing `aug_param`, images + cam2img + lidar2img.
    python tests/golden/make_augment_golden.py  ->  tests/golden/augment.npz"""
import copy
import random
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
CONF = {"reisze": [36, 45, 54], "crop": (0, 4, 80, 49), "H": 49, "W": 80, "rand_flip": True}


def inputs(seed, cams=2):
    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 255, (49, 80, 3)).astype(np.float32) for _ in range(cams)]
    cam2img = [np.eye(4) for _ in range(cams)]
    for k in cam2img:
        k[:3, :3] = np.array([[60.0, 0, 40], [0, 60.0, 24], [0, 0, 1]])
    lidar2cam = [np.eye(4) + 0.01 * rng.standard_normal((4, 4)) for _ in range(cams)]
    return imgs, cam2img, lidar2cam


SEEDS = [0, 1, 2, 5]


def main():
    import ref_import as R
    R.install_stubs()
    R._mod("mmdet.datasets"); R._mod("mmdet.datasets.builder", PIPELINES=R._Registry())
    A = R.load_file("ref_augmentation", R.PLUGIN / "datasets/pipelines/augmentation.py")
    out = {}
    for seed in SEEDS:
        random.seed(seed); np.random.seed(seed)
        aug = A.CropResizeFlipImage(data_aug_conf=CONF, training=True)
        aug_param = None
        for frame in range(2):
            imgs, cam2img, lidar2cam = inputs(seed * 10 + frame)
            res = dict(img=[i.copy() for i in imgs], cam2img=copy.deepcopy(cam2img), lidar2cam=copy.deepcopy(lidar2cam),
                       lidar2img=[None] * len(imgs))
            if aug_param is not None:
                res["aug_param"] = copy.deepcopy(aug_param)
            res = aug(res)
            aug_param = res["aug_param"]
            out[f"s{seed}_f{frame}_img"] = np.stack(res["img"])
            out[f"s{seed}_f{frame}_cam2img"] = np.stack(res["cam2img"])
            out[f"s{seed}_f{frame}_lidar2img"] = np.stack(res["lidar2img"])
        p = aug_param["CropResizeFlipImage_param"]
        out[f"s{seed}_param"] = np.array([p[0], p[1][0], p[1][1], float(p[3])])
    np.savez_compressed(HERE / "augment.npz", **out)


if __name__ == "__main__":
    main()
