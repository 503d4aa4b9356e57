"""Golden vectors for the head ray-march math from the REFERENCE's own ViDARHeadBase methods
(/root/reference imported with mmcv/mmdet/mmdet3d stubbed, see ref_import.py):
    python tests/golden/make_head_golden.py
Small volume (8 x 20 x 24) so the fixture stays small; ray_grid_num = 512 as in the configs."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(Path(__file__).parent))
import ref_import  # noqa: E402

head, e2e = ref_import.head_modules()
obj = object.__new__(head.ViDARHeadBase)
obj.__dict__.update(ray_grid_num=512, ray_grid_step=1.0, use_ce_loss=True, use_dist_loss=False,
                    use_dense_loss=True, dense_loss_weight=1.0, eval_within_grid=False,
                    loss_weight=np.array([[1.0], [0.5]]), _modules={}, _parameters={}, _buffers={})

torch.manual_seed(0)
Fn, Z, Y, X = 2, 8, 20, 24
pc_range = [-12.0, -10.0, -2.0, 12.0, 10.0, 2.0]          # 1 m voxels in x/y, 0.5 m in z
P = 90
# metric GT points [P,5] (x,y,z,intensity,frame): some outside the range, frames 0/1
pts = torch.cat([torch.rand(P, 1) * 28 - 14, torch.rand(P, 1) * 24 - 12, torch.rand(P, 1) * 5 - 2.5,
                 torch.rand(P, 1), torch.randint(0, Fn, (P, 1)).float()], 1)
gt_points = [pts]
origin_pts = torch.tensor([[[0.3, -0.2, 0.1], [1.5, 0.7, -0.3]]])         # [bs=1, F, 3] metric
# next_bev_preds [F, inter=1, bs=1, Y*X, Z]
bev_preds = torch.randn(Fn, 1, 1, Y * X, Z, requires_grad=True)
pred_dict = dict(next_bev_preds=bev_preds, valid_frames=[0, 1])

NOISE = {}
def fake_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
    g = -torch.empty_like(logits).exponential_(generator=torch.Generator().manual_seed(5)).log()
    NOISE["g"] = g
    idx = torch.softmax(logits + g, dim).max(dim, keepdim=True)[1]
    return torch.zeros_like(logits).scatter_(dim, idx, 1.0)
head.F.gumbel_softmax = fake_gumbel

loss = head.ViDARHeadBase.loss(obj, pred_dict, gt_points, 0, Y, X, pc_range, Fn,
                               batched_origin_points=origin_pts.clone())
total = loss["regularization.loss"] + 2.0 * loss["loss.dense_voxel"]
gsig, = torch.autograd.grad(total, bev_preds)

# intermediate pieces for kernel-level parity
(og, op, gg, gp, gti) = head.ViDARHeadBase._process_gt_points(
    obj, bev_preds.detach()[:, -1:], gt_points, origin_pts.clone(), [0, 1], 0, Fn, Y, X, pc_range)
sigma = bev_preds.detach()[:, 0].permute(1, 0, 3, 2).contiguous().view(1, Fn, Z, Y, X)
mask, feat, w, length = head.ViDARHeadBase._get_grid_features(
    obj, og, gg, gti, [sigma], obj.loss_weight, ray_grid_step=1.0)
ce = torch.nn.functional.cross_entropy(feat.transpose(1, 2).contiguous(),
                                       torch.zeros(1, feat.shape[1], dtype=torch.long), reduction="none")
decode = head.ViDARHeadBase.get_point_cloud_prediction(
    obj, dict(next_bev_preds=bev_preds.detach(), valid_frames=[0, 1]), gt_points, 0, Y, X, pc_range,
    batched_origin_points=origin_pts.clone())

out = Path(__file__).parent / "head_small.npz"
np.savez_compressed(
    out, bev_preds=bev_preds.detach().numpy(), gt_points=pts.numpy(), origin_pts=origin_pts.numpy(),
    pc_range=np.array(pc_range), loss_weight=obj.loss_weight,
    loss_ce=loss["regularization.loss"].detach().numpy(), loss_dense=loss["loss.dense_voxel"].detach().numpy(),
    grad_bev_preds=gsig.numpy(), noise=NOISE["g"].numpy(),
    origin_grids=og.numpy(), gt_grids=gg.numpy(), gt_tindex=gti.numpy(),
    feat=feat.numpy(), length=length.numpy(), ce=ce.numpy(), weight=w.numpy(),
    pred_pcd0=decode["pred_pcds"][0][0].numpy(), pred_pcd1=decode["pred_pcds"][0][1].numpy(),
    gt_pcd0=decode["gt_pcds"][0][0].numpy(), gt_pcd1=decode["gt_pcds"][0][1].numpy())
print(out, out.stat().st_size, {k: float(v) for k, v in loss.items()}, feat.shape, NOISE["g"].shape)
