"""ResNet101-DCNv2+FPN forward / forward+backward timing under MIOpen settings."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import vidar_amd.plugin as P
from vidar_amd.configs import get_config

cfg = get_config("vidar_1_8_nusc_1future", with_backbone=True)["model"]
bb = P.build_backbone(cfg["img_backbone"]).cuda()
neck = P.build_neck(cfg["img_neck"]).cuda()
x = torch.randn(6, 3, 928, 1600, device="cuda")


def run(tag, cl, bench, grad):
    torch.backends.cudnn.benchmark = bench
    xx = x.contiguous(memory_format=torch.channels_last) if cl else x
    m = bb.to(memory_format=torch.channels_last) if cl else bb.to(memory_format=torch.contiguous_format)
    n = neck.to(memory_format=torch.channels_last) if cl else neck.to(memory_format=torch.contiguous_format)
    def step():
        if grad:
            out = n(m(xx)); sum(o.sum() for o in out).backward()
        else:
            with torch.no_grad():
                n(m(xx))
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): step()
    torch.cuda.synchronize()
    print(f"{tag:40s} {(time.time()-t0)/3*1e3:8.1f} ms / 6 images", flush=True)

for grad in (False, True):
    for cl in (False, True):
        for bench in (False, True):
            run(f"grad={grad} channels_last={cl} benchmark={bench}", cl, bench, grad)
