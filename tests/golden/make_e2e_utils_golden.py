"""Golden for vidar_amd/plugin/utils/e2e_predictor_utils.py from the reference's own
bevformer/utils/e2e_predictor_utils.py: the pure grid / coordinate helpers, the chamfer wrappers (on
the reference's chamferdist python + its knn_cpu.cpp build) and the two autograd layers
DifferentiableVoxelRenderingLayer{,V2} wired to the reference's dvxlr / dvxlr_v2 kernels compiled for
the host (oracle/_ref/ref_dvxlr*.so).      python tests/golden/make_e2e_utils_golden.py"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(ROOT))
import ref_mmcv_functional as R  # noqa: E402

PC = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def inputs():
    g = torch.Generator().manual_seed(0)
    grids = torch.rand(2, 3, 50, 2, generator=g) * 1.4 - 0.2
    coords = (torch.rand(2, 3, 50, 3, generator=g) - 0.5) * torch.tensor([130.0, 130.0, 12.0])
    pts = (torch.rand(400, 3, generator=g) - 0.5) * torch.tensor([140.0, 140.0, 14.0])
    pred = (torch.rand(300, 3, generator=g) - 0.5) * torch.tensor([120.0, 120.0, 9.0])
    return grids, coords, pts, pred


def ray_case():
    from vidar_amd.synthetic import ray_set
    sigma, origin, points, tindex = ray_set(seed=9, N=1, T=2, rays_per_frame=40, pad=3, grid=(6, 20, 24),
                                            origin_jitter=3.0)
    return [torch.from_numpy(np.ascontiguousarray(a)) for a in (sigma, origin, points, tindex)]


def main():
    from oracle import build_ref
    e2e, _ = R.reference_eval_stack()
    grids, coords, pts, pred = inputs()
    out = {}
    out["grids_to_coords"] = e2e.bev_grids_to_coordinates(grids.clone(), PC).numpy()
    g2, m2 = e2e.bev_coords_to_grids(coords[..., :2].clone(), 20, 24, PC)
    out["coords_to_grids"], out["coords_to_grids_mask"] = g2.numpy(), m2.numpy()
    out["coords_to_voxel_grids"] = e2e.coords_to_voxel_grids(coords.clone(), 20, 24, 16, PC).numpy()
    for off in (0.5, 0.0):
        out[f"bev_grids_{off}"] = e2e.get_bev_grids(5, 7, bs=2, offset=off).numpy()
    out["bev_grids_3d"] = e2e.get_bev_grids_3d(4, 6, 3, bs=2).numpy()
    out["inside_mask"] = e2e.get_inside_mask(pts, PC).numpy()
    out["cd"] = np.float64(e2e.compute_chamfer_distance(pred, pts))
    out["cd_inner"] = np.float64(e2e.compute_chamfer_distance_inner(pred, pts, PC))
    out["cd_inner_empty"] = np.float64(e2e.compute_chamfer_distance_inner(pred + 1000.0, pts, PC))

    # autograd layers on the reference's own kernels (host build)
    e2e.dvxlr = build_ref.load("ref_dvxlr")
    e2e.dvxlr_v2 = build_ref.load("ref_dvxlr_v2")
    sigma, origin, points, tindex = ray_case()
    s = sigma.clone().requires_grad_(True)
    p, gdist = e2e.DifferentiableVoxelRenderingLayer.apply(s, origin, points, tindex)
    w = torch.randn(p.shape, generator=torch.Generator().manual_seed(2))
    (p * w).sum().backward()
    out.update(l1_pred=p.detach().numpy(), l1_gt=gdist.detach().numpy(), l1_w=w.numpy(), l1_grad=s.grad.numpy())
    s2 = sigma.clone().requires_grad_(True)
    reg = torch.rand(sigma.shape, generator=torch.Generator().manual_seed(4)).requires_grad_(True)
    p2, g2_, rp, ind = e2e.DifferentiableVoxelRenderingLayerV2.apply(s2, origin, points, tindex, reg)
    wr = torch.randn(rp.shape, generator=torch.Generator().manual_seed(3))
    ((p2 * w).sum() + (rp * wr * (ind >= 0)).sum()).backward()
    out.update(l2_pred=p2.detach().numpy(), l2_ray_pred=rp.detach().numpy(), l2_indicator=ind.detach().numpy(),
               l2_wr=wr.numpy(), l2_reg=reg.detach().numpy(), l2_grad=s2.grad.numpy(), l2_grad_reg=reg.grad.numpy())
    np.savez_compressed(HERE / "e2e_utils.npz", **out)
    print("wrote e2e_utils.npz", float(out["cd"]), float(out["cd_inner"]), float(np.abs(out["l1_grad"]).sum()),
          float(np.abs(out["l2_grad_reg"]).sum()))


if __name__ == "__main__":
    main()
