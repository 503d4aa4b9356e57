"""Weight states for measuring the step: `init` (what `init_weights` leaves) and `trained_like`.

The reference never trains from `init_weights` alone: it starts from a pretrained ResNet101-DCNv2
(`load_from='pretrained/r101_dcn_fcos3d_pretrain.pth'`, config vidar_1_8_nusc_3future.py:400), whose
`conv_offset` layers produce non-zero, per-pixel offsets and non-uniform masks, and a few iterations into
training `sampling_offsets.weight` / `attention_weights.weight` of every deformable attention (zero at init:
temporal_self_attention.py:110-125, spatial_cross_attention.py:230-246, vidar_decoder.py:355-371) make the
sampling pattern depend on the query.  At `init` every DCNv2 tap samples ON integer pixels and every query of
a head samples the same ring -- the cheapest access pattern the gather / scatter kernels can meet.

`apply_trained_like` puts the model into the harder, realistic regime without a checkpoint (there is no
network): it draws the data-dependent layers at random and then CALIBRATES them against the activations of one
forward pass of the given batch, layer by layer in execution order, so that
    DCNv2 offsets          ~ N(0, (dcn_offset_px)^2) pixels per tap, per pixel       (default 1.5 px)
    DCNv2 mask logits      ~ N(0, dcn_mask_logit^2)   (mask = sigmoid)                (default 1.0)
    MSDA sampling offsets  = init ring + N(0, (msda_offset_px)^2) per query           (default 1.5 px)
    MSDA attention logits  ~ N(0, msda_logit^2) per query (non-uniform softmax)       (default 1.0)
whatever the scale of the activations feeding them.  Everything else keeps its initialisation (no other kernel's
access pattern depends on weights).  A released `.pth` is loaded with `vidar_amd.checkpoint` instead when one is given.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

MODES = ("init", "trained_like")


def _dcn_packs(model):
    from .plugin.backbones import ModulatedDeformConv2dPack
    return [(n, m) for n, m in model.named_modules() if isinstance(m, ModulatedDeformConv2dPack)]


def _deform_attentions(model):
    return [(n, m) for n, m in model.named_modules()
            if isinstance(getattr(m, "sampling_offsets", None), torch.nn.Linear)
            and isinstance(getattr(m, "attention_weights", None), torch.nn.Linear)]


def _randn_like(p, gen):
    return torch.randn(p.shape, generator=gen, dtype=torch.float32).to(device=p.device, dtype=p.dtype)


@torch.no_grad()
def apply_trained_like(model, batch, seed=0, dcn_offset_px=1.5, dcn_mask_logit=1.0, msda_offset_px=1.5,
                       msda_logit=1.0):
    """In place.  `batch` = forward_train kwargs (one no-grad forward pass is run on it for the calibration).
    -> report dict: per layer kind, how many layers were calibrated and the standard deviations measured AFTER
    calibration on the same activations (they equal the targets up to rounding)."""
    gen = torch.Generator().manual_seed(seed)            # host generator: the draw does not depend on the device
    report = {"mode": "trained_like", "seed": seed,
              "targets": {"dcn_offset_px": dcn_offset_px, "dcn_mask_logit": dcn_mask_logit,
                          "msda_offset_px": msda_offset_px, "msda_logit": msda_logit},
              "dcn_layers": 0, "msda_layers": 0, "uncalibrated": []}
    hooks, todo = [], {}

    def scale_rows(w, rows, x_out, target):
        sd = float(x_out.float().std())
        if not (sd > 0.0) or sd != sd:
            return None
        w.data[rows] *= target / sd              # .data: the model's forward may switch autograd back on around the hook
        return sd

    # --- DCNv2: conv_offset [27, C, 3, 3] -> channels 0..17 offsets (pixels), 18..26 mask logits -----------------
    for name, pack in _dcn_packs(model):
        co = pack.conv_offset
        co.weight.copy_(_randn_like(co.weight, gen))
        k2 = pack.k * pack.k
        b = torch.zeros_like(co.bias)
        b[:2 * k2] = _randn_like(b[:2 * k2], gen) * (dcn_offset_px / 3.0)       # a small constant drift per tap
        b[2 * k2:] = _randn_like(b[2 * k2:], gen) * (dcn_mask_logit / 2.0)
        co.bias.copy_(b)
        todo[name] = "dcn"

        def pre(mod, args, name=name, k2=k2):
            if todo.pop(name, None) is None:
                return
            co = mod.conv_offset
            with torch.no_grad():
                out = F.conv2d(args[0].float(), co.weight.float(), None, co.stride, co.padding, co.dilation)
            a = scale_rows(co.weight, slice(0, 2 * k2), out[:, :2 * k2], dcn_offset_px)
            m = scale_rows(co.weight, slice(2 * k2, 3 * k2), out[:, 2 * k2:], dcn_mask_logit)
            if a is None or m is None:
                report["uncalibrated"].append(name)
            else:
                report["dcn_layers"] += 1
        hooks.append(pack.register_forward_pre_hook(pre))

    # --- deformable attention: Linear outputs in pixels (sampling_offsets) and logits (attention_weights) --------
    for name, att in _deform_attentions(model):
        so, aw = att.sampling_offsets, att.attention_weights
        so.weight.copy_(_randn_like(so.weight, gen))                           # the bias keeps the init ring
        aw.weight.copy_(_randn_like(aw.weight, gen))
        aw.bias.copy_(_randn_like(aw.bias, gen) * (msda_logit / 2.0))
        for lin, target, key in ((so, msda_offset_px, name + ".sampling_offsets"),
                                 (aw, msda_logit, name + ".attention_weights")):
            todo[key] = "msda"

            def pre(mod, args, key=key, target=target):
                if todo.pop(key, None) is None:
                    return
                with torch.no_grad():
                    out = F.linear(args[0].float(), mod.weight.float())
                if scale_rows(mod.weight, slice(None), out, target) is None:
                    report["uncalibrated"].append(key)
                elif key.endswith("sampling_offsets"):
                    report["msda_layers"] += 1
            hooks.append(lin.register_forward_pre_hook(pre))

    was_training = model.training
    # the calibration products (F.linear / F.conv2d without the layers' biases) are shapes of their own: keep TunableOp from
    # spending the warm-up tuning them
    tuning = None
    try:
        import torch.cuda.tunable as tn
        if torch.cuda.is_available() and tn.is_enabled() and tn.tuning_is_enabled():
            tuning = tn
            tn.tuning_enable(False)
    except (ImportError, RuntimeError):
        tuning = None
    try:
        model(return_loss=True, **batch)
    finally:
        if tuning is not None:
            tuning.tuning_enable(True)
        for h in hooks:
            h.remove()
        model.train(was_training)
    report["uncalibrated"] += sorted(todo)               # layers the forward pass never reached
    return report


def prepare(model, batch, mode="init", checkpoint=None, **kw):
    """bench / tools entry: mode in MODES, or a checkpoint path (mmcv `.pth` layout, vidar_amd.checkpoint)."""
    if checkpoint:
        from .checkpoint import load_checkpoint
        _, missing, unexpected = load_checkpoint(model, checkpoint)
        return {"mode": "checkpoint", "file": str(checkpoint), "missing_keys": len(missing),
                "unexpected_keys": len(unexpected)}
    if mode == "init":
        return {"mode": "init"}
    if mode == "trained_like":
        return apply_trained_like(model, batch, **kw)
    raise ValueError(f"weights mode {mode!r}: expected one of {MODES} or a checkpoint path")
