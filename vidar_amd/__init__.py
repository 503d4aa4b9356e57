"""vidar_amd -- MI355X-native hot path of ViDAR (BEV-encode -> latent-render -> chamfer).

Host side mirrors the reference's plugin / extension surface; all heavy lifting is done by
hand-written gfx950 HIP kernels in libvidar_hip.so behind the C ABI of include/vidar_hip.h.
"""
__version__ = "0.1.0"
