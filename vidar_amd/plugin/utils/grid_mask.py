"""GridMask -- the training-time augmentation the reference applies to the camera images on the GPU
(projects/mmdet3d_plugin/models/utils/grid_mask.py:69-124; call sites detectors/vidar.py:139-140 images,
:145-148 backbone features, :156-159 FPN features, :289-294 previous BEV).

Same random draws in the same order from numpy's global generator (rand, randint(2, h), randint(d) x 2,
randint(rotate)), so a seeded run masks exactly what the reference masks.  The reference builds the
1.5h x 1.5w mask on the host with python loops, rotates it with PIL and uploads it every call; here the
un-rotated mask (rotate == 1 in every released config -> angle 0) is assembled ON DEVICE from the four
scalars as a row predicate OR a column predicate, so nothing but a few bytes crosses PCIe.  A non-zero
rotation angle falls back to the reference's host construction (PIL nearest-neighbour rotate)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class GridMask(nn.Module):
    def __init__(self, use_h, use_w, rotate=1, offset=False, ratio=0.5, mode=0, prob=1.0):
        super().__init__()
        self.use_h, self.use_w, self.rotate, self.offset = use_h, use_w, rotate, offset
        self.ratio, self.mode, self.st_prob, self.prob = ratio, mode, prob, prob

    def set_prob(self, epoch, max_epoch):
        self.prob = self.st_prob * epoch / max_epoch

    @staticmethod
    def _stripes(n_full, n, d, st, length, device):
        """bool [n]: positions (after the centre crop of the 1.5x canvas) that a stripe zeroes."""
        r = torch.arange(n, device=device) + (n_full - n) // 2
        k = r - st
        return (k >= 0) & (k % d < length) & (torch.div(k, d, rounding_mode="floor") < n_full // d) & (r < n_full)

    def forward(self, x):
        if np.random.rand() > self.prob or not self.training:
            return x
        n, c, h, w = x.size()
        hh, ww = int(1.5 * h), int(1.5 * w)
        d = np.random.randint(2, h)
        length = min(max(int(d * self.ratio + 0.5), 1), d - 1)
        st_h = np.random.randint(d)
        st_w = np.random.randint(d)
        r = np.random.randint(self.rotate)
        if r == 0:
            zero = torch.zeros((h, w), dtype=torch.bool, device=x.device)
            if self.use_h:
                zero = zero | self._stripes(hh, h, d, st_h, length, x.device)[:, None]
            if self.use_w:
                zero = zero | self._stripes(ww, w, d, st_w, length, x.device)[None, :]
            mask = (~zero).to(x.dtype)
        else:                                        # host construction, as the reference
            from PIL import Image
            m = np.ones((hh, ww), np.float32)
            if self.use_h:
                for i in range(hh // d):
                    s = d * i + st_h
                    m[s:min(s + length, hh), :] = 0
            if self.use_w:
                for i in range(ww // d):
                    s = d * i + st_w
                    m[:, s:min(s + length, ww)] = 0
            m = np.asarray(Image.fromarray(np.uint8(m)).rotate(r))
            m = m[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w]
            mask = torch.from_numpy(np.ascontiguousarray(m)).to(device=x.device, dtype=x.dtype)
        if self.mode == 1:
            mask = 1 - mask
        if self.offset:
            off = torch.from_numpy(2 * (np.random.rand(h, w) - 0.5)).to(device=x.device, dtype=x.dtype)
            return x * mask + off * (1 - mask)
        return x * mask
