"""Host-side mirror of the reference's `projects/mmdet3d_plugin` registry surface for the hot path
(same registered type names, constructor kwargs and parameter names).  Importing this package
registers every module, like `importlib.import_module('projects.mmdet3d_plugin')` does in the
reference (tools/train.py:113-137)."""
from . import bricks, backbones  # noqa: F401
from .modules import (temporal_self_attention, spatial_cross_attention, encoder, transformer,  # noqa: F401
                      vidar_decoder, vidar_transformer)
from .modules.ray_operations import latent_rendering  # noqa: F401
from .dense_heads import vidar_bevformer_head, vidar_head_base, vidar_head_v1  # noqa: F401
from .detectors import vidar  # noqa: F401
from .registry import *  # noqa: F401,F403
from .config import Config  # noqa: F401
