"""CPU: oracle/latent_render.py restatement vs golden vectors produced by the reference's own
LatentRendering module (tests/golden/make_latent_render_golden.py)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import latent_render as LR

GOLD = Path(__file__).parent / "golden"
CASES = {"14x14_s1_sigmoid": (1.0, "sigmoid"), "10x16_s05_sigmoid": (0.5, "sigmoid"),
         "9x9_s1_exp": (1.0, "exp")}


def load(name):
    g = np.load(GOLD / f"latent_render_{name}.npz")
    t = {k: torch.from_numpy(g[k]) for k in g.files}
    return t


@pytest.mark.parametrize("name", list(CASES))
def test_restatement_matches_reference_module(name):
    step, act = CASES[name]
    t = load(name)
    embed = t["embed"].clone().requires_grad_(True)
    ps = [t["p_unsup_raymarching_head.0.weight"], t["p_unsup_raymarching_head.0.bias"],
          t["p_lora_a.weight"], t["p_lora_a.bias"], t["p_lora_b.weight"], t["p_lora_b.bias"]]
    ps = [p.clone().requires_grad_(True) for p in ps]
    out = LR.forward(embed, *ps, grid_num=256, grid_step=step, act=act)
    torch.testing.assert_close(out, t["out"], rtol=1e-5, atol=1e-6)
    grads = torch.autograd.grad((out * t["gout"]).sum(), [embed, *ps])
    torch.testing.assert_close(grads[0], t["grad_embed"], rtol=1e-4, atol=1e-6)
    for g, k in zip(grads[1:], ["unsup_raymarching_head.0.weight", "unsup_raymarching_head.0.bias",
                                "lora_a.weight", "lora_a.bias", "lora_b.weight", "lora_b.bias"]):
        torch.testing.assert_close(g, t["g_" + k], rtol=1e-4, atol=1e-5)
