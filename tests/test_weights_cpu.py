"""CPU: the `trained_like` weight state of bench.py (vidar_amd/weights.py) -- every data-dependent layer of the step
(DCNv2 conv_offset, deformable-attention sampling_offsets / attention_weights) leaves its zero initialisation and is
calibrated to the stated pixel / logit scales on the batch's own activations; ops routed to the CPU oracle."""
import numpy as np
import torch
import torch.nn.functional as F

from test_plugin_cpu import _small_batch


def _tiny_image_batch():
    from vidar_amd.configs import get_config
    from vidar_amd.synthetic import make_sample
    cfg = get_config("vidar_1_8_nusc_1future", bev_h=24, bev_w=24, with_backbone=True)
    hw = (64, 96)
    metas, gt = make_sample(0, rays_per_frame=100, future_frames=cfg["future_frames"], num_cams=cfg["num_cams"], img_hw=hw)
    for m in metas:
        m["img_shape"] = [(hw[0], hw[1], 3)] * cfg["num_cams"]
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 5, cfg["num_cams"], 3, hw[0], hw[1], generator=g)
    return cfg, dict(img_metas=[metas], gt_points=[torch.from_numpy(gt)], img=img)


def test_trained_like_calibrates_every_data_dependent_layer():
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd import weights as W
    torch.manual_seed(0); np.random.seed(0)
    cfg, batch = _tiny_image_batch()
    model = T.build_model(cfg).train()
    packs = W._dcn_packs(model)
    atts = W._deform_attentions(model)
    assert len(packs) == 26 and len(atts) >= 12           # ResNet101 stages 3-4 (23 + 3 DCNv2 blocks); 6 x (TSA + SCA) + decoder
    assert all(float(p.conv_offset.weight.detach().abs().max()) == 0.0 for _, p in packs)        # the init the review flagged
    seen = {}

    def probe(name, k2):
        def hook(mod, args):
            if name in seen:
                return
            co = mod.conv_offset
            o = F.conv2d(args[0], co.weight, None, co.stride, co.padding, co.dilation)
            seen[name] = (float(o[:, :2 * k2].std()), float(o[:, 2 * k2:].std()))
        return hook
    with cpu_ops.patched():
        rep = W.prepare(model, batch, "trained_like", seed=3)
        assert rep["uncalibrated"] == [] and rep["dcn_layers"] == len(packs) and rep["msda_layers"] == len(atts)
        hs = [p.register_forward_pre_hook(probe(n, p.k * p.k)) for n, p in packs]
        lin = {}
        for n, a in atts:
            def lin_probe(m, args, n=n):
                lin.setdefault(n, float(F.linear(args[0], m.weight).std()))
            hs.append(a.sampling_offsets.register_forward_pre_hook(lin_probe))
        with torch.no_grad():
            model(return_loss=True, **batch)
        for h in hs:
            h.remove()
        # the same batch again: the measured spreads are the targets (layer by layer in execution order)
        for n, (off, mask) in seen.items():
            assert abs(off - 1.5) < 0.05 and abs(mask - 1.0) < 0.05, (n, off, mask)
        assert all(abs(v - 1.5) < 0.05 for v in lin.values()), lin
        # and the step still trains: finite loss, a gradient for every trainable parameter
        opt = T.build_optimizer(model)
        loss, _ = T.train_step(model, opt, batch)
    assert torch.isfinite(loss)
    assert not [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    # the ring bias of the sampling offsets is kept (offsets = init ring + per-query part)
    so = atts[0][1].sampling_offsets
    assert float(so.bias.abs().max()) > 0.5


def test_init_mode_is_a_no_op_and_bad_modes_raise():
    import pytest
    from vidar_amd import train as T
    from vidar_amd import weights as W
    torch.manual_seed(0)
    cfg, batch = _small_batch("vidar_1_8_nusc_1future")
    model = T.build_model(cfg)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    assert W.prepare(model, batch, "init") == {"mode": "init"}
    assert all(torch.equal(v, model.state_dict()[k]) for k, v in before.items())
    with pytest.raises(ValueError):
        W.prepare(model, batch, "pretrained")


def test_trained_like_is_deterministic_in_its_seed():
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd import weights as W
    outs = []
    for _ in range(2):
        torch.manual_seed(0); np.random.seed(0)
        cfg, batch = _small_batch("vidar_1_8_nusc_1future")
        model = T.build_model(cfg).eval()
        with cpu_ops.patched():
            W.apply_trained_like(model, batch, seed=5)
        outs.append(torch.cat([p.detach().flatten()[::13] for n, p in model.named_parameters() if "sampling_offsets" in n]))
    assert torch.equal(outs[0], outs[1])
