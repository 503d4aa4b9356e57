"""BEVFormerEncoder / CustomBEVFormerEncoder / BEVFormerLayerV2 -- names, kwargs and semantics of
projects/mmdet3d_plugin/bevformer/modules/encoder.py:27-253 and encoder_v2.py:27-209.
TransformerLayerSequence protocol (mmcv, third party): `transformerlayers` dict deep-copied
`num_layers` times into `self.layers`."""
from __future__ import annotations

import copy

import numpy as np
import torch
import torch.nn as nn

from ..bricks import can_fuse_norm
from ..utils.host import const_tensor, to_device_async
from ..registry import TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, build_transformer_layer
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer
from .ray_operations.latent_rendering import LatentRendering
from .spatial_cross_attention import visible_query_index


class ScaPlan:
    """What one encoder pass needs about one frame's cameras: the projected pillar anchors, their validity
    and the visible-query index of SpatialCrossAttention -- built for all frames of a step by one call of
    vidar_sca_plan_f32 (BEVFormerEncoder.plan_frames) with one host read of the list lengths."""
    __slots__ = ("ref_cam", "bev_mask", "index", "slot_of", "valid_u8", "stride", "ref_re")

    def __init__(self, ref_cam, bev_mask, index, slot_of=None, valid_u8=None, stride=0):
        self.ref_cam, self.bev_mask, self.index = ref_cam, bev_mask, index
        self.slot_of, self.valid_u8, self.stride = slot_of, valid_u8, stride     # inverse index (HIP rebatch)
        self.ref_re = None                       # rebatched reference points, filled by the first SCA layer

    def __deepcopy__(self, memo):          # img_metas are deep-copied by the detector (vidar.py:286)
        return self


class TransformerLayerSequence(nn.Module):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, (list, tuple)) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList([build_transformer_layer(dict(c)) for c in transformerlayers])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


class BEVFormerEncoder(TransformerLayerSequence):
    def __init__(self, *args, pc_range=None, num_points_in_pillar=4, return_intermediate=False,
                 dataset_type="nuscenes", latent_rendering_lid=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.num_points_in_pillar = num_points_in_pillar
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.latent_rendering_lid = latent_rendering_lid

    @staticmethod
    def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim="3d", bs=1, device="cuda",
                             dtype=torch.float):
        """pillar anchors [bs, D, H*W, 3] ('3d') or cell centres [bs, H*W, 1, 2] ('2d'), all in
        [0,1] (encoder.py:54-92)."""
        xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device) / W
        ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device) / H
        if dim == "3d":
            D = num_points_in_pillar
            zs = torch.linspace(0.5, Z - 0.5, D, dtype=dtype, device=device) / Z
            ref = torch.stack((xs.view(1, 1, W).expand(D, H, W), ys.view(1, H, 1).expand(D, H, W),
                               zs.view(D, 1, 1).expand(D, H, W)), -1).reshape(D, H * W, 3)
            return ref[None].repeat(bs, 1, 1, 1)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        ref = torch.stack((gx.reshape(-1), gy.reshape(-1)), -1)
        return ref[None].repeat(bs, 1, 1).unsqueeze(2)

    def point_sampling(self, reference_points, pc_range, img_metas):
        """project pillar anchors into every camera (encoder.py:96-156): fp32 (TF32 never used on
        this path), depth clamp 1e-5, strict in-image test, normalised by camera 0's padded shape."""
        lidar2img = reference_points.new_tensor(np.asarray([m["lidar2img"] for m in img_metas]),
                                                dtype=torch.float32)           # [B, N, 4, 4]
        pts = reference_points.float().clone()
        for a in range(3):
            pts[..., a:a + 1] = pts[..., a:a + 1] * (pc_range[a + 3] - pc_range[a]) + pc_range[a]
        pts = torch.cat((pts, torch.ones_like(pts[..., :1])), -1)              # [B, D, Q, 4]
        # cam[d, b, n, q, :] = lidar2img[b, n] @ pts[b, d, q]  (no 61 MB repeat of the matrices)
        cam = torch.einsum("bnij,bdqj->dbnqi", lidar2img, pts)
        eps = 1e-5
        mask = cam[..., 2:3] > eps
        xy = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
        xy[..., 0] /= img_metas[0]["img_shape"][0][1]
        xy[..., 1] /= img_metas[0]["img_shape"][0][0]
        mask = (mask & (xy[..., 1:2] > 0.0) & (xy[..., 1:2] < 1.0) & (xy[..., 0:1] < 1.0)
                & (xy[..., 0:1] > 0.0))
        mask = torch.nan_to_num(mask)
        return xy.permute(2, 1, 3, 0, 4), mask.permute(2, 1, 3, 0, 4).squeeze(-1)

    def _cached_points(self, bev_h, bev_w, device, dtype):
        """pillar anchors [1, D, Q, 3] and cell centres [1, Q, 1, 2]: constants of the model, built once"""
        key = (bev_h, bev_w, str(device), dtype)
        cache = self.__dict__.setdefault("_ref_cache", {})
        if key not in cache:
            cache[key] = (self.get_reference_points(bev_h, bev_w, self.pc_range[5] - self.pc_range[2],
                                                    self.num_points_in_pillar, dim="3d", bs=1, device=device,
                                                    dtype=dtype),
                          self.get_reference_points(bev_h, bev_w, dim="2d", bs=1, device=device, dtype=dtype))
        return cache[key]

    def plan_frames(self, metas_per_frame, bev_h, bev_w, device, dtype=torch.float32):
        """point_sampling + visible-query index for several frames in ONE kernel call and one host read.
        metas_per_frame: list over frames of the per-sample img_metas list.  -> list of ScaPlan.
        (HIP path; on CPU tensors -- the oracle-routed tests -- the torch formulation below is used.)"""
        if device.type != "cuda":
            return [None] * len(metas_per_frame)
        import ctypes
        from ..._lib import lib, check, ptr, stream_of
        F, B = len(metas_per_frame), len(metas_per_frame[0])
        l2i = np.asarray([[m["lidar2img"] for m in metas] for metas in metas_per_frame], dtype=np.float32)
        N = l2i.shape[2]
        ref_3d = self._cached_points(bev_h, bev_w, device, dtype)[0].float().contiguous()
        D, Q = ref_3d.shape[1], ref_3d.shape[2]
        l2i_d = to_device_async(l2i, device)
        ref_cam = torch.empty((F, N, B, Q, D, 2), device=device)
        mask = torch.empty((F, N, B, Q, D), device=device, dtype=torch.uint8)
        count = torch.empty((F, B, Q), device=device)
        idx = torch.empty((F, N, Q), device=device, dtype=torch.int64)
        valid = torch.empty((F, N, Q), device=device, dtype=torch.uint8)
        lens = torch.empty((F, N), device=device, dtype=torch.int32)
        slot_of = torch.empty((F, N, Q), device=device, dtype=torch.int32)
        shape0 = metas_per_frame[0][0]["img_shape"][0]
        rng = (ctypes.c_float * 6)(*[float(v) for v in self.pc_range])
        check(lib().vidar_sca_plan_f32(ptr(ref_3d), ptr(l2i_d), ptr(ref_cam), ptr(mask), ptr(count), ptr(idx),
                                       ptr(valid), ptr(lens), ptr(slot_of), rng, ctypes.c_float(float(shape0[0])),
                                       ctypes.c_float(float(shape0[1])), F, B, N, Q, D, stream_of(ref_cam)),
              "sca_plan")
        max_len = lens.max(dim=1).values.tolist()            # the one host read of the step
        # round the padded length up to a multiple of 256: the GEMMs of the cross attention then see a small set
        # of shapes (tuned once, stable allocator blocks) for <= 3 % more masked slots
        max_len = [min(Q, (m + 255) // 256 * 256) if m > 0 else 0 for m in max_len]
        return [ScaPlan(ref_cam[f], mask[f].bool(),
                        (idx[f, :, :max_len[f]], valid[f, :, :max_len[f]].bool(), count[f]),
                        slot_of=slot_of[f], valid_u8=valid[f], stride=Q)
                for f in range(F)]

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, prev_bev=None,
                shift=0., **kwargs):
        output = bev_query
        intermediate = []
        bs = bev_query.size(1)
        ref_3d, ref_2d = self._cached_points(bev_h, bev_w, bev_query.device, bev_query.dtype)
        ref_3d, ref_2d = ref_3d.repeat(bs, 1, 1, 1), ref_2d.repeat(bs, 1, 1, 1)
        img_metas = kwargs["img_metas"]
        plan = img_metas[0].get("_sca_plan")
        if plan is None:                                  # not planned by the detector: plan this frame alone
            plan = self.plan_frames([img_metas], bev_h, bev_w, bev_query.device, bev_query.dtype)[0]
        if plan is not None:
            reference_points_cam, bev_mask, sca_index = plan.ref_cam, plan.bev_mask, plan.index
        else:                                             # CPU tensors (oracle-routed tests)
            reference_points_cam, bev_mask = self.point_sampling(ref_3d, self.pc_range, img_metas)
            sca_index = visible_query_index(bev_mask)    # once per pass instead of once per layer
        shift_ref_2d = ref_2d.clone() + shift[:, None, None, :]
        bev_query = bev_query.permute(1, 0, 2)
        # [Q, bs, C] view of a [bs, C, H, W] map -> [bs, Q, C]: made contiguous ONCE per pass; the six layers add it
        # to the queries twice each, and an add against the channel-strided view runs at a third of the rate
        bev_pos = bev_pos.permute(1, 0, 2).contiguous()
        bs, len_bev, num_bev_level, _ = ref_2d.shape
        if prev_bev is not None:
            prev_bev = prev_bev.permute(1, 0, 2)
            prev_bev = torch.stack([prev_bev, bev_query], 1).reshape(bs * 2, len_bev, -1)
            hybird_ref_2d = torch.stack([shift_ref_2d, ref_2d], 1).reshape(bs * 2, len_bev, num_bev_level, 2)
        else:
            hybird_ref_2d = torch.stack([ref_2d, ref_2d], 1).reshape(bs * 2, len_bev, num_bev_level, 2)
        # TemporalSelfAttention concatenates `value[:bs]` to its query (temporal_self_attention.py:183; "value" = this
        # stacked tensor, the same one for all six layers).  Sliced ONCE here instead of once per layer: six slice
        # backwards each zero-fill a [2 bs, Q, C] gradient, copy their half in and add it to the tensor's gradient
        value_head = prev_bev[:bs] if prev_bev is not None else None
        for lid, layer in enumerate(self.layers):
            output = layer(bev_query, key, value, *args, bev_pos=bev_pos, ref_2d=hybird_ref_2d,
                           ref_3d=ref_3d, bev_h=bev_h, bev_w=bev_w, spatial_shapes=spatial_shapes,
                           level_start_index=level_start_index,
                           reference_points_cam=reference_points_cam, bev_mask=bev_mask,
                           prev_bev=prev_bev, sca_index=sca_index, sca_plan=plan, value_head=value_head, **kwargs)
            bev_query = output
            if self.latent_rendering_lid is not None:
                if prev_bev is not None and lid in self.latent_rendering_lid:
                    # reference: prev_bev[:bs] (encoder.py:244-245) -- the history rows only for bs == 1,
                    # the layout is (b0 prev, b0 cur, b1 prev, ...): take every sample's history row
                    hist = prev_bev.view(bs, 2, len_bev, -1)[:, 0]
                    prev_bev = torch.stack([hist, bev_query], 1).reshape(bs * 2, len_bev, -1)
                    value_head = prev_bev[:bs]
            if self.return_intermediate:
                intermediate.append(output)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class CustomBEVFormerEncoder(BEVFormerEncoder):
    def __init__(self, keep_idx=(2,), *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.keep_idx = keep_idx
        for lid, layer in enumerate(self.layers):
            if lid not in self.keep_idx and getattr(layer, "latent_render", None) is not None:
                del layer.latent_render
                layer.operation_order = ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")

    def forward(self, *args, **kwargs):
        default = self.return_intermediate
        self.return_intermediate = kwargs.pop("return_intermediate", self.return_intermediate)
        try:
            return super().forward(*args, **kwargs)
        finally:
            self.return_intermediate = default


@TRANSFORMER_LAYER.register_module()
class BEVFormerLayerV2(MyCustomBaseTransformerLayer):
    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2,
                 latent_render=None, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        self.fp16_enabled = False
        if latent_render is not None:
            self.latent_render = LatentRendering(**latent_render)

    def forward(self, query, key=None, value=None, bev_pos=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, ref_2d=None,
                ref_3d=None, bev_h=None, bev_w=None, reference_points_cam=None, mask=None,
                spatial_shapes=None, level_start_index=None, prev_bev=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        self_shapes = const_tensor([[bev_h, bev_w]], query.device, torch.int64)
        self_lsi = const_tensor([0], query.device, torch.int64)
        ops = self.operation_order
        skip = False
        for k, layer in enumerate(ops):
            if skip:                      # this norm was fused into the block in front of it
                skip = False
                continue
            # post-norm layers: "block ; norm" = LayerNorm(dropout(block) + identity) -> one fused pass
            fuse = None
            if not self.pre_norm and layer in ("self_attn", "cross_attn", "ffn") and k + 1 < len(ops) \
                    and ops[k + 1] == "norm" and can_fuse_norm(self.norms[norm_index], query):
                fuse = self.norms[norm_index]
                norm_index += 1
                skip = True
            if layer == "self_attn":
                query = self.attentions[attn_index](
                    query, prev_bev, prev_bev, identity if self.pre_norm else None, query_pos=bev_pos,
                    key_pos=bev_pos, key_padding_mask=query_key_padding_mask, reference_points=ref_2d,
                    spatial_shapes=self_shapes, level_start_index=self_lsi, fuse_norm=fuse, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "norm":
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == "cross_attn":
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, reference_points=ref_3d, reference_points_cam=reference_points_cam,
                    mask=mask, key_padding_mask=key_padding_mask, spatial_shapes=spatial_shapes,
                    level_start_index=level_start_index, fuse_norm=fuse, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "latent_render":
                bs, token_num, embed_dim = query.shape
                query = self.latent_render(query.view(bs, bev_h, bev_w, embed_dim)).view(
                    bs, token_num, embed_dim)
            elif layer == "ffn":
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None, fuse_norm=fuse)
                ffn_index += 1
        return query
