import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import vidar_amd.plugin as P
from vidar_amd.configs import get_config
from torch.profiler import profile, ProfilerActivity
cfg = get_config("vidar_1_8_nusc_1future", with_backbone=True)["model"]
bb = P.build_backbone(cfg["img_backbone"]).cuda(); neck = P.build_neck(cfg["img_neck"]).cuda()
x = torch.randn(2, 3, 464, 800, device="cuda")
for _ in range(2):
    with torch.no_grad(): neck(bb(x))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    with torch.no_grad(): neck(bb(x))
evs = [e for e in prof.events() if e.name in ("aten::copy_", "aten::contiguous", "aten::clone")]
from collections import Counter
c = Counter()
for e in evs:
    st = [s for s in (e.stack or []) if "vidar_amd" in s or "tools/" in s]
    c[(e.name, st[0] if st else "?", str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(12): print(v, k)
y = torch.nn.functional.conv2d(torch.randn(2, 64, 116, 200, device="cuda"), torch.randn(256, 64, 1, 1, device="cuda"))
print("conv out contiguous:", y.is_contiguous(), y.stride())
