"""Golden vectors for LatentRendering from the REFERENCE module itself (imported from
/root/reference with mmcv stubbed, see ref_import.py).  Run in the build container:
    python tests/golden/make_latent_render_golden.py"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(Path(__file__).parent))
import ref_import  # noqa: E402

m = ref_import.latent_rendering_module()
for name, (H, W, step, act) in {"14x14_s1_sigmoid": (14, 14, 1.0, "sigmoid"),
                                "10x16_s05_sigmoid": (10, 16, 0.5, "sigmoid"),
                                "9x9_s1_exp": (9, 9, 1.0, "exp")}.items():
    torch.manual_seed(0)
    mod = m.LatentRendering(embed_dims=256, pred_height=16, num_pred_fcs=0, grid_step=step,
                            grid_num=256, reduction=16, act=act)
    embed = torch.randn(1, H, W, 256, requires_grad=True)
    out = mod(embed)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad((out * gout).sum(), [embed, *params.values()])
    np.savez_compressed(Path(__file__).parent / f"latent_render_{name}.npz",
                        embed=embed.detach().numpy(), out=out.detach().numpy(), gout=gout.numpy(),
                        grad_embed=grads[0].numpy(),
                        **{"p_" + k: v.detach().numpy() for k, v in params.items()},
                        **{"g_" + k: g.numpy() for k, g in zip(params, grads[1:])})
    print(name, out.shape, float(out.abs().mean()), [k for k in params])
