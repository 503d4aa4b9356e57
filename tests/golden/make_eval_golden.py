"""Golden values for the ray-error metrics from the REFERENCE's utils/eval_utils.py (numpy host
code; `chamferdist` stubbed with a CPU nearest-neighbour of identical semantics):
    python tests/golden/make_eval_golden.py"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(Path(__file__).parent))
import ref_import  # noqa: E402
from oracle import chamfer as C  # noqa: E402


class CpuChamfer(torch.nn.Module):
    def forward(self, src, tgt, bidirectional=False, reverse=False, reduction="mean"):
        assert reverse and not bidirectional
        idx, d = C.knn_points_idx(tgt.numpy(), src.numpy())
        return torch.from_numpy(d[..., 0]).sum(1).mean(), (torch.from_numpy(d[..., 0]), torch.from_numpy(idx[..., 0]))


sys.modules["chamferdist"] = types.SimpleNamespace(ChamferDistance=CpuChamfer)
ref = ref_import.load_file("ref_eval_utils", ref_import.PLUGIN / "bevformer/utils/eval_utils.py")
rng = np.random.default_rng(0)
out = {}
for name, origin in {"origin_inside": np.array([1.0, -2.0, 0.3]), "origin_outside": np.array([80.0, 5.0, 1.0])}.items():
    gt = np.concatenate([rng.uniform(-60, 60, (300, 3)) * [1, 1, 0.05], rng.uniform(-120, 120, (80, 3)) * [1, 1, 0.08]])
    pred = gt + rng.normal(0, 1.5, gt.shape)
    l1, ar = ref.compute_ray_errors(pred.copy(), gt.copy(), origin.copy(), torch.device("cpu"))
    o, p, inv = ref.clamp(gt.copy(), origin.copy(), return_invalid_mask=True)
    out.update({f"{name}_gt": gt, f"{name}_pred": pred, f"{name}_origin": origin, f"{name}_l1": l1,
                f"{name}_absrel": ar, f"{name}_clamp_o": o, f"{name}_clamp_p": p, f"{name}_invalid": inv})
    print(name, l1, ar, int(inv.sum()))
np.savez_compressed(Path(__file__).parent / "eval_ray_errors.npz", **out)
