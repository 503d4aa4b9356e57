from __future__ import annotations

import math

import torch

from ..bricks import constant_init, xavier_init


def init_deformable_offsets(sampling_offsets, num_heads, n_groups, num_points):
    """DETR-style ring initialisation of the offset bias shared by every deformable attention of
    the reference (temporal_self_attention.py:101-117, spatial_cross_attention.py:252-267,
    vidar_decoder.py:361-376): head h points at angle 2*pi*h/H, point i at radius i+1."""
    constant_init(sampling_offsets, 0.)
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(num_heads, 1, 1, 2).repeat(
        1, n_groups, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    sampling_offsets.bias.data = grid.view(-1)
