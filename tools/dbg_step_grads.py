import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import cpu_ops
from vidar_amd import train as T
from test_plugin_cpu import _small_batch
from test_step_gpu import _to
name = sys.argv[1] if len(sys.argv) > 1 else "vidar_1_8_nusc_1future"
torch.manual_seed(0); np.random.seed(0)
cfg, batch = _small_batch(name)
model = T.build_model(cfg)
model.random_drop_prev_rate = 0.0
g_ = torch.Generator().manual_seed(11)
for n_, p_ in model.named_parameters():
    if n_.endswith("sampling_offsets.bias"):
        p_.data += torch.randn(p_.shape, generator=g_) * 0.3
noise = -torch.empty(20000, 512).exponential_(generator=torch.Generator().manual_seed(3)).log()
model.future_pred_head.gumbel_noise_fn = lambda R, K: noise[:R].to(next(model.parameters()).device)
model.train(); model.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.Dropout) else None)
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = [p for p in model.parameters() if p.requires_grad]
which = sys.argv[2] if len(sys.argv) > 2 else "all"
def total(d):
    if which == "ce": return sum(v for k, v in d.items() if "regularization" in k)
    if which == "dense": return sum(v for k, v in d.items() if "dense" in k)
    return sum(d.values())
with cpu_ops.patched():
    ref = model(return_loss=True, **batch)
    rg = torch.autograd.grad(total(ref), params, allow_unused=True)
model.cuda()
out = model(return_loss=True, **_to(batch, "cuda"))
g = torch.autograd.grad(total(out), params, allow_unused=True)
rows = []
for n, a, b in zip(names, g, rg):
    if a is None or b is None: continue
    e = float((a.cpu() - b).norm()); d = float(b.norm())
    rows.append((e / (d + 1e-12), e, d, n))
for r in sorted(rows, reverse=True)[:14]: print(f"{r[0]:.3e} err {r[1]:.3e} ref {r[2]:.3e} {r[3]}")
print("losses", {k: (round(float(out[k]), 5), round(float(ref[k]), 5)) for k in ref})
