"""Host-side image augmentation of the training pipeline (config :308-311), for `ViDARSequenceDataset(augment=)`.

  CropResizeFlipImage   projects/mmdet3d_plugin/datasets/pipelines/augmentation.py:10-203 -- one crop / resize /
                        flip drawn per SAMPLE and replayed on every frame of its queue through `aug_param`
                        (nuscenes_vidar_dataset_template.py:116-124); rescales cam2img and rebuilds lidar2img; the
                        detector flips the FPN features back when the flip bit is set (detectors/vidar.py:123-126).
                        Pinned against the reference class (tests/golden/make_augment_golden.py).
  PhotoMetricDistortion pipelines/transform_3d.py:99-190 -- brightness / contrast / saturation / hue / channel swap
                        with the reference's draw order.  The BGR<->HSV conversion is mmcv's (cv2, float32: H in
                        degrees, S in [0,1], V = max channel) [3P, restated: cv2 is not installed here] -> UNPINNED.
Both work on a record dict(img [cams,3,H,W] float tensor BEFORE normalisation, img_metas) -- i.e. they are meant to
run between `load_images(..., normalise=False)` and the normalisation; `make_train_augment` wires that order."""
from __future__ import annotations

import copy
import random

import numpy as np
import torch

IDA_AUG_CONF = {"reisze": [720, 765, 810, 855, 900, 945, 990, 1035, 1080], "crop": (0, 0, 1600, 900),
                "H": 900, "W": 1600, "rand_flip": True}                           # config :41-47 (sic: "reisze")


class CropResizeFlipImage:
    def __init__(self, data_aug_conf=None, training=True):
        self.conf = dict(data_aug_conf or IDA_AUG_CONF)
        self.training = training

    def sample(self, aug_param):
        """(resize, resize_dims, crop, flip): replayed from `aug_param` when the queue already drew it (:179-203)"""
        if "CropResizeFlipImage_param" in aug_param:
            return aug_param["CropResizeFlipImage_param"]
        crop = self.conf["crop"]
        resized_h = random.choice(self.conf["reisze"])
        if not self.training:
            assert len(self.conf["reisze"]) == 1
        resized_w = resized_h / (crop[3] - crop[1]) * (crop[2] - crop[0])
        resize = resized_h / (crop[3] - crop[1])
        flip = bool(self.training and self.conf["rand_flip"] and np.random.choice([0, 1]))
        aug_param["CropResizeFlipImage_param"] = (resize, (int(resized_w), int(resized_h)), crop, flip)
        return aug_param["CropResizeFlipImage_param"]

    def __call__(self, imgs, meta, aug_param):
        """imgs: list of HxWx3 float arrays (BGR, 0..255); meta: needs cam2img + lidar2cam.  -> new list;
        meta['cam2img'] / ['lidar2img'] updated in place (:88-89, :137-138)."""
        from PIL import Image
        resize, resize_dims, crop, flip = self.sample(aug_param)
        out = []
        ida = np.eye(3)
        ida[:2, :2] *= resize
        ida[:2, 2] = -np.array(crop[:2]) * resize
        for i, img in enumerate(imgs):
            im = Image.fromarray(np.uint8(img)).crop(crop).resize(resize_dims)
            if flip:
                im = im.transpose(method=Image.FLIP_LEFT_RIGHT)
            out.append(np.array(im).astype(np.float32))
            meta["cam2img"][i][:3, :3] = np.matmul(ida, meta["cam2img"][i][:3, :3])
        meta["lidar2img"] = [np.matmul(meta["cam2img"][i], meta["lidar2cam"][i]) for i in range(len(meta["lidar2cam"]))]
        return out


def bgr2hsv(img):
    """[3P] cv2.COLOR_BGR2HSV on float32: V = max, S = (V - min) / V (0 where V == 0), H in degrees [0, 360)"""
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = img.max(-1)
    d = v - img.min(-1)
    s = np.where(v > 0, d / np.where(v > 0, v, 1), 0)
    dd = np.where(d > 0, d, 1)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2 + (b - r) / dd, 4 + (r - g) / dd)) * 60
    h = np.where(d > 0, h, 0)
    h = np.where(h < 0, h + 360, h)
    return np.stack([h, s, v], -1).astype(np.float32)


def hsv2bgr(img):
    """[3P] cv2.COLOR_HSV2BGR on float32"""
    h, s, v = img[..., 0] / 60.0, img[..., 1], img[..., 2]
    i = np.floor(h).astype(np.int32) % 6
    f = h - np.floor(h)
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([b, g, r], -1).astype(np.float32)


class PhotoMetricDistortionMultiViewImage:
    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.brightness_delta = brightness_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range
        self.hue_delta = hue_delta

    @staticmethod
    def _coin_uniform(lo, hi):
        """the reference's draw pattern: a fair coin first, the magnitude only when the coin says yes"""
        return np.random.uniform(lo, hi) if np.random.randint(2) else None

    def _distort(self, img):
        img = np.array(img, dtype=np.float32)
        shift = self._coin_uniform(-self.brightness_delta, self.brightness_delta)
        if shift is not None:
            img += shift
        contrast_first = np.random.randint(2) == 1               # "mode": contrast before or after the HSV stage
        gain = self._coin_uniform(self.contrast_lower, self.contrast_upper) if contrast_first else None
        if gain is not None:
            img *= gain
        hsv = bgr2hsv(img)
        sat = self._coin_uniform(self.saturation_lower, self.saturation_upper)
        if sat is not None:
            hsv[..., 1] *= sat
        turn = self._coin_uniform(-self.hue_delta, self.hue_delta)
        if turn is not None:
            hue = hsv[..., 0] + turn
            hue[hue > 360] -= 360
            hue[hue < 0] += 360
            hsv[..., 0] = hue
        img = hsv2bgr(hsv)
        gain = None if contrast_first else self._coin_uniform(self.contrast_lower, self.contrast_upper)
        if gain is not None:
            img *= gain
        if np.random.randint(2):
            img = img[..., np.random.permutation(3)]
        return img

    def __call__(self, imgs):
        """every image of the rig draws its own distortion, in order (numpy's global generator, like the reference's
        `from numpy import random`)"""
        return [self._distort(img) for img in imgs]
