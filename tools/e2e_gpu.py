"""One end-to-end training step of vidar_1_8_nusc_1future on synthetic inputs (GPU)."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from vidar_amd import train as T
from vidar_amd.configs import get_config
from vidar_amd.synthetic import make_sample, fpn_features

name = sys.argv[1] if len(sys.argv) > 1 else "vidar_1_8_nusc_1future"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = get_config(name)
torch.manual_seed(0); np.random.seed(0)
model = T.build_model(cfg).cuda().train()
opt = T.build_optimizer(model)
metas, gt = make_sample(0, queue_length=4, future_frames=cfg["future_frames"])
feats = fpn_features(0, 5, device="cuda")
batch = dict(img_metas=[metas], gt_points=[torch.from_numpy(gt).cuda()], img_feats=feats)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.time()
    total, losses = T.train_step(model, opt, batch)
    torch.cuda.synchronize()
    print(i, f"{(time.time()-t0)*1e3:.1f} ms", float(total), {k: round(float(v), 4) for k, v in losses.items()}, flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 1e9)
