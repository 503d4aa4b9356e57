"""GPU: ONE full-size forward pass in one piece -- images [1, 2, 6, 3, 928, 1600] -> ResNet101-DCNv2 + FPN -> BEV encoder
at 200 x 200 (1 frame of history + the current frame) -> head -> ray CE / gumbel render / chamfer -- on the HIP path
against the same pass with every op routed to the CPU oracle on the host (forward only, no autograd: it fits in a few GB).
The ops are covered one by one at full size elsewhere (test_fullsize_parity_gpu.py, test_msda_gpu.py, ...); this is the
piece the reduced-BEV step tests (test_step_gpu.py) cannot see: the full-size step's own shapes meeting each other."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_full_size_forward_matches_the_oracle_routed_forward():
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd import weights as W
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import make_sample
    torch.manual_seed(0); np.random.seed(0)
    cfg = get_config("vidar_1_8_nusc_1future", with_backbone=True)
    cfg["model"]["use_grid_mask"] = False
    # one frame of history instead of four (the host pass must stay at minutes): the head then predicts 1 history frame +
    # the current one + 1 future, everything else -- image size, BEV 200 x 200, 30 000 rays per frame -- is the config's own
    head = cfg["model"]["future_pred_head"]
    head["history_queue_length"], head["pred_history_frame_num"], head["per_frame_loss_weight"] = 1, 1, (0.6, 1.0, 1.2)
    metas, gt = make_sample(0, queue_length=1, future_frames=cfg["future_frames"], rays_per_frame=30000,
                            num_cams=cfg["num_cams"], img_hw=cfg["img_hw"])
    g = torch.Generator().manual_seed(2)
    img = torch.randn(1, 2, cfg["num_cams"], 3, *cfg["img_hw"], generator=g)
    import copy
    # (own metas per pass: the detector parks per-step device tensors -- the SCA plan -- in the meta dictionaries)
    batch = dict(img=img, img_metas=[copy.deepcopy(metas)], gt_points=[torch.from_numpy(gt)])
    model = T.build_model(cfg)
    for m in model.modules():
        if hasattr(m, "random_drop_prev_rate"):
            m.random_drop_prev_rate = 0.0
    model.train()
    model.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.Dropout) else None)
    noise = -torch.empty(10000, 512).exponential_(generator=torch.Generator().manual_seed(3)).log()
    model.future_pred_head.gumbel_noise_fn = lambda R, K: noise[:R].to(next(model.parameters()).device)
    model.cuda()
    dev = dict(img=img.cuda(), img_metas=[metas], gt_points=[torch.from_numpy(gt).cuda()])
    # trained-like weights (per-pixel DCNv2 offsets, per-query attention offsets): the harder sampling pattern, and
    # off the bilinear kinks of the initial integer rings
    rep = W.apply_trained_like(model, dict(dev, img_metas=[copy.deepcopy(metas)]), seed=1)
    assert not rep["uncalibrated"]
    with torch.no_grad():
        out = {k: float(v) for k, v in model(return_loss=True, **dev).items()}
    model.cpu()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad(), cpu_ops.patched():
        ref = {k: float(v) for k, v in model(return_loss=True, **batch).items()}
    assert set(out) == set(ref) and len(ref) == 6
    for k in ref:
        assert np.isfinite(ref[k])
        np.testing.assert_allclose(out[k], ref[k], rtol=5e-3, atol=1e-5, err_msg=k)
