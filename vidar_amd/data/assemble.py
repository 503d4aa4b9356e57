"""Turn per-frame records into the training / test sample the detector consumes.

Mirrors (behaviour, not code):
  * NuScenesViDARDatasetTemplate.__init__ usable-index scan and _prepare_data_info index lists
    (projects/mmdet3d_plugin/datasets/nuscenes_vidar_dataset_template.py:44-68, :101-135);
  * NuScenesViDARDatasetV1.union2one / _mask_points
    (projects/mmdet3d_plugin/datasets/nuscenes_vidar_dataset_v1.py:22-203).
A record is a plain dict(img=Tensor[cams,3,H,W], points=array/Tensor[P,>=4], img_metas=dict, optional
aug_param); img_metas carries scene_token, can_bus[18], ego2global_{translation,rotation},
lidar2ego_{translation,rotation} (rotations as (w,x,y,z) quaternions) plus whatever the model reads
(lidar2img, img_shape, lidar2global_rotation, sample_idx).  The reference wraps the same fields in
mmcv DataContainers; the collate step that unwraps them is the loader's business.

[3P] `transform_matrix` restates nuscenes.utils.geometry_utils.transform_matrix with pyquaternion's
rotation matrix (both third party, not vendored by the reference)."""
from __future__ import annotations

import copy
from typing import List, Optional, Sequence

import numpy as np
import torch


def _rotation_matrix(q) -> np.ndarray:
    w, x, y, z = np.asarray(q, np.float64) / np.linalg.norm(np.asarray(q, np.float64))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def transform_matrix(translation, rotation, inverse: bool = False) -> np.ndarray:
    """4x4 homogeneous transform of (translation, unit quaternion (w,x,y,z)); column-vector form."""
    tm = np.eye(4)
    rot = _rotation_matrix(rotation)
    t = np.asarray(translation, np.float64)
    if inverse:
        tm[:3, :3] = rot.T
        tm[:3, 3] = rot.T.dot(-t)
    else:
        tm[:3, :3] = rot
        tm[:3, 3] = t
    return tm


def frame_meta_from_info(info: dict, dataset: str = "nuscenes", data_root: str = "") -> dict:
    """info dict (entry of the mmdet3d-style `*_infos_*.pkl`) -> the per-frame meta the model and
    union2one read: lidar2img per camera, lidar2global_rotation, can_bus with the pose patched in.
    dataset="nuscenes": CustomNuScenesDataset.get_data_info (datasets/nuscenes_dataset.py:153-227) plus
    the lidar2ego / cam2img fields of the template's override (nuscenes_vidar_dataset_template.py:70-80);
    dataset="nuplan" (OpenScene): NuPlanViDARDatasetTemplate.get_data_info
    (nuplan_vidar_dataset_template.py:48-118) -- paths joined with data_root, sample_prev/next, no sweeps."""
    import os
    if dataset not in ("nuscenes", "nuplan"):
        raise ValueError(f"unknown dataset flavour {dataset!r}")
    nuplan = dataset == "nuplan"
    lidar2ego_r = _rotation_matrix(info["lidar2ego_rotation"])
    ego2global_r = _rotation_matrix(info["ego2global_rotation"])
    meta = dict(sample_idx=info["token"],
                pts_filename=os.path.join(data_root, info["lidar_path"]) if nuplan else info["lidar_path"],
                sweeps=[] if nuplan else info["sweeps"],
                ego2global_translation=info["ego2global_translation"],
                ego2global_rotation=info["ego2global_rotation"],
                lidar2global_rotation=ego2global_r @ lidar2ego_r,
                prev_idx=info["sample_prev" if nuplan else "prev"], next_idx=info["sample_next" if nuplan else "next"],
                scene_token=info["scene_token"], can_bus=info["can_bus"], frame_idx=info["frame_idx"],
                timestamp=info["timestamp"] / 1e6, lidar2ego_translation=info["lidar2ego_translation"],
                lidar2ego_rotation=info["lidar2ego_rotation"])
    paths, lidar2img, intrinsics, lidar2cam = [], [], [], []
    for cam in info["cams"].values():
        paths.append(os.path.join(data_root, cam["data_path"]) if nuplan else cam["data_path"])
        r = np.linalg.inv(cam["sensor2lidar_rotation"])
        t = cam["sensor2lidar_translation"] @ r.T
        rt = np.eye(4)
        rt[:3, :3] = r.T
        rt[3, :3] = -t
        k = cam["cam_intrinsic"]
        viewpad = np.eye(4)
        viewpad[:k.shape[0], :k.shape[1]] = k
        lidar2img.append(viewpad @ rt.T)
        intrinsics.append(viewpad)
        lidar2cam.append(rt.T)
    meta.update(img_filename=paths, lidar2img=lidar2img, lidar2cam=lidar2cam, cam2img=intrinsics)
    if not nuplan:
        meta["cam_intrinsic"] = intrinsics
    # overwrite the can-bus pose with the ego pose; yaw in [0, 360) degrees / radians
    can_bus = meta["can_bus"]
    can_bus[:3] = info["ego2global_translation"]
    can_bus[3:7] = info["ego2global_rotation"]
    v = ego2global_r @ np.array([1.0, 0.0, 0.0])
    angle = np.arctan2(v[1], v[0]) / np.pi * 180          # [3P] nuscenes quaternion_yaw
    if angle < 0:
        angle += 360
    can_bus[-2] = angle / 180 * np.pi
    can_bus[-1] = angle
    return meta


def usable_indices(data_infos: Sequence[dict], future_length: int, queue_length: int, test_mode: bool,
                   load_frame_interval: Optional[int] = None) -> List[int]:
    """Frames with enough history (test mode only: the 4d-occ protocol) and `future_length` future
    frames inside the same scene (template :44-68)."""
    last_scene, last_frame, out = None, -1, []
    need_prev = queue_length if test_mode else 0
    for index, info in enumerate(data_infos):
        if last_scene != info["scene_token"]:
            last_scene, last_frame = info["scene_token"], -1
        last_frame += 1
        if last_frame >= need_prev:
            tgt = index + future_length
            if tgt >= len(data_infos):
                break
            if last_scene != data_infos[tgt]["scene_token"]:
                continue
            out.append(index)
    return out[::load_frame_interval] if load_frame_interval is not None else out


def frame_index_lists(index: int, queue_length: int, future_length: int, rand_interval: int, n: int):
    """-> (previous frame indices incl. `index`, future frame indices starting at `index`), clamped to
    the dataset (template :101-135; a negative interval walks the chain backwards)."""
    prev = sorted(range(index - queue_length * rand_interval, index, rand_interval))
    if rand_interval < 0:
        prev = prev[::-1]
    prev.append(index)
    fut = sorted(range(index, index + (future_length + 1) * rand_interval, rand_interval))
    if rand_interval < 0:
        fut = fut[::-1]
    clamp = lambda i: min(max(0, i), n - 1)
    return [clamp(i) for i in prev], [clamp(i) for i in fut]


def _mask_ego(pts, ego_mask):
    inside = ((ego_mask[0] <= pts[:, 0]) & (ego_mask[2] >= pts[:, 0]) &
              (ego_mask[1] <= pts[:, 1]) & (ego_mask[3] >= pts[:, 1]))
    return pts[~inside]


def _np(p):
    return p.detach().cpu().numpy() if isinstance(p, torch.Tensor) else np.asarray(p)


def _lidar_chain(meta):
    e2g = transform_matrix(meta["ego2global_translation"], meta["ego2global_rotation"])
    g2e = transform_matrix(meta["ego2global_translation"], meta["ego2global_rotation"], inverse=True)
    l2e = transform_matrix(meta["lidar2ego_translation"], meta["lidar2ego_rotation"])
    e2l = transform_matrix(meta["lidar2ego_translation"], meta["lidar2ego_rotation"], inverse=True)
    return e2g, g2e, l2e, e2l


def union2one(previous_queue: List[dict], future_queue: List[dict], future_length: int, ego_mask=None):
    """-> dict(img [T,cams,3,H,W], img_metas {frame: meta}, gt_points [sum P, C]) or None when the
    scene ends before `future_length` futures (v1 :38-203).  The last column of gt_points becomes the
    frame index inside previous_queue[:-1] + future_queue; all 4x4 matrices are in the reference's
    row-vector convention (points @ M)."""
    ref_meta = previous_queue[-1]["img_metas"]
    r_e2g, r_g2e, r_l2e, r_e2l = _lidar_chain(ref_meta)

    queue = previous_queue[:-1] + future_queue
    pts_list = [_np(each["points"]) for each in queue]
    if ego_mask is not None:
        pts_list = [_mask_ego(p, ego_mask) for p in pts_list]
    cur2ref, ref2cur, total_pts = [], [], []
    for i, each in enumerate(queue):
        cur = pts_list[i].copy()
        cur[:, -1] = i
        total_pts.append(cur)
        c_e2g, c_g2e, c_l2e, c_e2l = _lidar_chain(each["img_metas"])
        cur2ref.append(c_l2e.T @ c_e2g.T @ r_g2e.T @ r_e2l.T)
        ref2cur.append(r_l2e.T @ r_e2g.T @ c_g2e.T @ c_e2l.T)

    # history: can_bus becomes the motion since the previous frame of the same scene
    metas_map = {}
    prev_scene = prev_pos = prev_angle = None
    for i, each in enumerate(previous_queue):
        meta = metas_map[i] = each["img_metas"]
        if "aug_param" in each:
            meta["aug_param"] = each["aug_param"]
        can_bus = copy.deepcopy(meta["can_bus"])
        if meta["scene_token"] != prev_scene:
            meta["prev_bev_exists"] = False
            prev_scene = meta["scene_token"]
            prev_pos, prev_angle = copy.deepcopy(can_bus[:3]), copy.deepcopy(can_bus[-1])
            can_bus[:3] = 0
            can_bus[-1] = 0
        else:
            meta["prev_bev_exists"] = True
            pos, angle = copy.deepcopy(can_bus[:3]), copy.deepcopy(can_bus[-1])
            can_bus[:3] = pos - prev_pos
            can_bus[-1] = angle - prev_angle
            prev_pos, prev_angle = pos, angle
        meta["can_bus"] = can_bus
        meta["ref_lidar_to_cur_lidar"] = ref2cur[i]

    # futures: stop at the first frame of another scene
    scene = ref_meta["scene_token"]
    ref_can_bus = None
    future_can_bus, future2ref, ref2future = [], [], []
    off = len(previous_queue) - 1
    for i, each in enumerate(future_queue):
        meta = each["img_metas"]
        if meta["scene_token"] != scene:
            break
        future2ref.append(cur2ref[i + off])
        ref2future.append(ref2cur[i + off])
        can_bus = copy.deepcopy(meta["can_bus"])
        if i == 0:
            can_bus[:3] = 0
            can_bus[-1] = 0
        else:
            pos = np.array([0, 0, 0, 1]).reshape(1, 4) @ future2ref[-1] @ ref2future[-2]
            can_bus[-1] = can_bus[-1] - ref_can_bus[-1]
            can_bus[:3] = pos[:, :3]
        future_can_bus.append(can_bus)
        ref_can_bus = copy.deepcopy(meta["can_bus"])

    last = metas_map[len(previous_queue) - 1]
    last["future_can_bus"] = np.array(future_can_bus)
    last["future2ref_lidar_transform"] = np.array(future2ref)
    last["ref2future_lidar_transform"] = np.array(ref2future)
    last["total_cur2ref_lidar_transform"] = np.array(cur2ref)
    last["total_ref2cur_lidar_transform"] = np.array(ref2cur)
    if len(future_can_bus) < 1 + future_length:
        return None
    out = {k: v for k, v in previous_queue[-1].items() if k not in ("points", "aug_param")}
    out["img"] = torch.stack([each["img"] for each in previous_queue])
    out["img_metas"] = metas_map
    out["gt_points"] = torch.from_numpy(np.concatenate(total_pts, 0))
    return out
