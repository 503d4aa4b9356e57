"""nuScenes / OpenScene info-pkl reader feeding `assemble.union2one` -- the data format on the input side
of the hot path.

What the reference does with mmdet3d's Custom3DDataset + a pipeline of registered transforms
(datasets/nuscenes_dataset.py:134-227, nuscenes_vidar_dataset_template.py:44-135,
nuscenes_vidar_dataset_v1.py:38-203; config :279-330), restated as plain functions on numpy:
  load_infos            Custom3DDataset.load_annotations (sorted by timestamp, load_interval)     [3P mmdet3d]
  load_points_file      LoadPointsFromFile(load_dim=5, use_dim=5)                                  [3P mmdet3d]
  load_multi_sweeps     CustomLoadPointsFromMultiSweeps (datasets/pipelines/loading.py:10-205)
  voxel_subsample       CustomVoxelBasedPointSampler (loading.py:225-241) on mmdet3d's VoxelGenerator
                        (first point of every occupied voxel, in order of first appearance)         [3P mmdet3d]
  load_images           LoadMultiViewImageFromFiles(to_float32) + NormalizeMultiviewImage +
                        PadMultiViewImage(size_divisor=32) (pipelines/transform_3d.py:8-95)         [3P mmcv imread]
  TrainAugment          PhotoMetricDistortionMultiViewImage + CropResizeFlipImage of the training pipeline
                        (config :308-311; vidar_amd/data/augment.py), enabled with `ViDARSequenceDataset(augment=True)`
[3P] pieces follow the published behaviour of mmcv 1.4.0 / mmdet3d 0.17.1 (not vendored by the reference)."""
from __future__ import annotations

import copy
import pickle
from pathlib import Path
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .assemble import frame_index_lists, frame_meta_from_info, union2one, usable_indices

IMG_NORM = dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False)     # config :50-51
PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def load_infos(ann_file, load_interval: int = 1):
    """-> (infos sorted by timestamp, every load_interval-th; metadata dict)"""
    with open(ann_file, "rb") as f:
        data = pickle.load(f)
    infos = list(sorted(data["infos"], key=lambda e: e["timestamp"]))[::load_interval]
    return infos, data.get("metadata", {})


def load_points_file(path, load_dim: int = 5) -> np.ndarray:
    p = str(path)
    pts = np.load(p) if p.endswith(".npy") else np.fromfile(p, dtype=np.float32)
    return np.copy(pts).reshape(-1, load_dim)


def load_pcd_file(path, fields=("x", "y", "z", "intensity", "ring", "lidar_info")) -> np.ndarray:
    """binary `.pcd` cloud of the OpenScene / nuPlan logs -> float32 [N, 6] (x, y, z, intensity, ring, lidar_info):
    LoadNuPlanPointsFromFile = PointCloud.parse_from_file(...).to_pcd_bin2().T
    (datasets/pipelines/nuplan_loading.py:73-181, :193-203).  Header keys as in the PCD format; only DATA binary
    with COUNT 1 per field is accepted, like the reference."""
    kinds = {"I": "int", "U": "uint", "F": "float"}
    with open(path, "rb") as f:
        hdr = {}
        while True:
            line = f.readline().decode("utf8").strip()
            if not line:
                raise RuntimeError(f"{path}: truncated .pcd header")
            if line.startswith("#"):
                continue
            cols = line.split()
            hdr[cols[0].lower()] = cols[1:] if len(cols) > 2 else cols[1]
            if cols[0].lower() == "data":
                break
        as_list = lambda v: v if isinstance(v, list) else [v]
        names, sizes, types, counts = (as_list(hdr[k]) for k in ("fields", "size", "type", "count"))
        if any(int(c) != 1 for c in counts):
            raise RuntimeError('"count" has to be 1')
        if not len(names) == len(sizes) == len(types) == len(counts):
            raise RuntimeError("fields/size/type/count field number are inconsistent")
        if hdr["data"] != "binary":
            raise RuntimeError(f'Un-supported data foramt: {hdr["data"]}. "binary" is expected.')
        row = np.dtype([(n, getattr(np, kinds[t] + str(int(sz) * 8))) for n, sz, t in zip(names, sizes, types)])
        need = row.itemsize * int(hdr["points"])
        buf = f.read(need)                                   # trailing garbage after the points is ignored
        if len(buf) != need:
            raise RuntimeError(f"Incomplete pointcloud stream: {need} bytes expected, {len(buf)} got")
    pts = np.frombuffer(buf, row)
    return np.stack([np.asarray(pts[k], dtype=np.float32) for k in fields], 1)


def remove_close(points: np.ndarray, radius: float = 1.0, ego_mask=None) -> np.ndarray:
    """drop |x|<r & |y|<r (loading.py:76-97), then the ego-vehicle box (:190-215; inclusive bounds)"""
    close = (np.abs(points[:, 0]) < radius) & (np.abs(points[:, 1]) < radius)
    points = points[~close]
    if ego_mask is not None:
        ego = ((ego_mask[0] <= points[:, 0]) & (ego_mask[2] >= points[:, 0])
               & (ego_mask[1] <= points[:, 1]) & (ego_mask[3] >= points[:, 1]))
        points = points[~ego]
    return points


def select_sweeps(sweeps: Sequence[dict], ts: float, sweeps_num: int, test_mode=False, random_select=True):
    """loading.py:99-116"""
    if len(sweeps) <= sweeps_num:
        return np.arange(len(sweeps))
    if test_mode:
        return np.arange(sweeps_num)
    if random_select:
        return np.random.choice(len(sweeps), sweeps_num, replace=False)
    gap = np.abs(np.array([s["timestamp"] for s in sweeps]) / 1e6 - ts)
    return np.argsort(gap)[:sweeps_num]


def load_multi_sweeps(points: np.ndarray, sweeps: Sequence[dict], ts: float, sweeps_num=2,
                      use_dim=(0, 1, 2, 3, 4), pad_empty_sweeps=True, remove_close_pts=True, ego_mask=None,
                      hard_sweeps_timestamp=0, random_select=False, test_mode=False, load_dim=5,
                      loader: Callable = load_points_file) -> np.ndarray:
    """key frame (time channel 0, NOT filtered) + `sweeps_num` sweeps moved into the key frame's lidar frame
    (loading.py:118-168), time channel overwritten by `hard_sweeps_timestamp` when given (:217-223)."""
    points = np.array(points, dtype=np.float32, copy=True)
    points[:, 4] = 0
    parts = [points]
    if pad_empty_sweeps and len(sweeps) == 0:
        for _ in range(sweeps_num):
            parts.append(remove_close(points, ego_mask=ego_mask) if remove_close_pts else points)
    else:
        for idx in select_sweeps(sweeps, ts, sweeps_num, test_mode, random_select):
            sweep = sweeps[idx]
            p = loader(sweep["data_path"], load_dim)
            if remove_close_pts:
                p = remove_close(p, ego_mask=ego_mask)
            p[:, :3] = p[:, :3] @ np.asarray(sweep["sensor2lidar_rotation"]).T
            p[:, :3] += np.asarray(sweep["sensor2lidar_translation"])
            p[:, 4] = ts - sweep["timestamp"] / 1e6
            parts.append(p.astype(np.float32))
    out = np.concatenate(parts, 0)[:, list(use_dim)]
    if hard_sweeps_timestamp is not None:
        out[:, 4] = hard_sweeps_timestamp
    return out


def voxel_subsample(points: np.ndarray, voxel_size=(1.0, 1.0, 1.0), point_cloud_range=PC_RANGE,
                    max_voxels: int = 50000) -> np.ndarray:
    """first point of every occupied voxel, voxels in order of first appearance, points outside the range
    dropped, at most `max_voxels` voxels -- mmdet3d VoxelGenerator(max_num_points=1) as used through
    CustomVoxelBasedPointSampler._sample_points (returns the un-padded voxel array, loading.py:226-241)."""
    rng = np.asarray(point_cloud_range, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    c = np.floor((points[:, :3] - rng[:3]) / vs).astype(np.int64)
    ok = ((c >= 0) & (c < grid)).all(1)
    key = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]
    idx = np.nonzero(ok)[0]
    _, first = np.unique(key[idx], return_index=True)
    keep = np.sort(idx[first])[:max_voxels]
    return points[keep]


def voxel_point_sampler(points: np.ndarray, voxel_size=(1.0, 1.0, 1.0), point_cloud_range=PC_RANGE,
                        max_voxels: int = 50000, time_dim: int = 4) -> np.ndarray:
    """`CustomVoxelBasedPointSampler.__call__` = mmdet3d v0.17.1 `VoxelBasedPointSampler.__call__` ([3P], restated
    from memory, unpinned: mmdet3d is not vendored) around the reference's `_sample_points` (loading.py:226-241):
    split into current-sweep points (time channel == 0) and previous-sweep points, SHUFFLE both with
    `np.random.shuffle`, voxel-sample the current sweep (the released configs give no `prev_sweep_cfg`, so the
    previous sweeps are dropped).  With `hard_sweeps_timestamp=0` every point carries time 0, the previous set is
    empty and mmdet3d aliases it to the current array -- which is therefore shuffled twice, in place.  So the point
    kept per voxel, the subset kept beyond `max_voxels` and the output order are random, and the numpy stream
    advances by two shuffles per frame."""
    cur_flag = points[:, time_dim] == 0
    cur = points[cur_flag]
    prev = points[~cur_flag]
    if prev.shape[0] == 0:
        prev = cur                                  # same array object, as in mmdet3d
    np.random.shuffle(cur)
    np.random.shuffle(prev)
    return voxel_subsample(cur, voxel_size, point_cloud_range, max_voxels)


def load_raw_images(paths: Sequence) -> list:
    """-> list of float32 [H, W, 3] arrays, channel order BGR like mmcv.imread (cv2), values 0..255
    (LoadMultiViewImageFromFiles(to_float32=True))"""
    from PIL import Image
    out = []
    for p in paths:
        if str(p).endswith(".npy"):
            out.append(np.load(p).astype(np.float32))               # already BGR float (test fixtures)
        else:
            out.append(np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"), dtype=np.float32)[..., ::-1]))
    return out


def normalise_pad(imgs: Sequence, mean=IMG_NORM["mean"], std=IMG_NORM["std"], to_rgb=IMG_NORM["to_rgb"],
                  size_divisor: int = 32, scale=None):
    """NormalizeMultiviewImage [+ RandomScaleImageMultiViewImage(scales=[scale])] + PadMultiViewImage(size_divisor)
    -> (float32 [cams, 3, H_pad, W_pad], padded shape); `to_rgb` flips the channels before normalising like
    mmcv.imnormalize; zero padding at bottom / right.  `scale` (OpenScene: 2/3, transform_3d.py:295-327) resizes to
    (int(h*scale), int(w*scale)) with bilinear interpolation at pixel centres ([3P] mmcv.imresize = cv2.INTER_LINEAR;
    torch's bilinear, align_corners=False, no antialias, is the same sampling rule) -- the caller scales lidar2img."""
    out = []
    for a in imgs:
        if to_rgb:
            a = a[..., ::-1]
        a = (a - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
        if scale is not None:
            hs, ws = int(a.shape[0] * scale), int(a.shape[1] * scale)
            t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))[None]
            a = torch.nn.functional.interpolate(t, size=(hs, ws), mode="bilinear", align_corners=False)[0] \
                .permute(1, 2, 0).numpy()
        h, w = a.shape[:2]
        H = (h + size_divisor - 1) // size_divisor * size_divisor
        W = (w + size_divisor - 1) // size_divisor * size_divisor
        pad = np.zeros((H, W, 3), np.float32)
        pad[:h, :w] = a
        out.append(pad)
    arr = np.stack(out)                                              # [cams, H, W, 3]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(0, 3, 1, 2))), tuple(arr.shape[1:])


def load_images(paths: Sequence, mean=IMG_NORM["mean"], std=IMG_NORM["std"], to_rgb=IMG_NORM["to_rgb"],
                size_divisor: int = 32):
    """load + normalise + pad (the test pipeline, config :319-330)"""
    return normalise_pad(load_raw_images(paths), mean, std, to_rgb, size_divisor)


class TrainAugment:
    """PhotoMetricDistortionMultiViewImage then CropResizeFlipImage (config :308-311), `aug_param` carried from
    frame to frame of a queue so that every frame of a sample gets the same crop / resize / flip."""

    def __init__(self, data_aug_conf=None, photometric=True, crop_resize_flip=True):
        from .augment import CropResizeFlipImage, PhotoMetricDistortionMultiViewImage
        self.photo = PhotoMetricDistortionMultiViewImage() if photometric else None
        self.crop = CropResizeFlipImage(data_aug_conf, training=True) if crop_resize_flip else None

    def __call__(self, imgs, meta, aug_param):
        if self.photo is not None:
            imgs = self.photo(imgs)
        return self.crop(imgs, meta, aug_param) if self.crop is not None else imgs


class ViDARSequenceDataset:
    """`dataset[i]` -> dict(img [T, cams, 3, H, W], img_metas {t: meta}, gt_points [P, 5]): the forward_train /
    forward_test kwargs of one sample (collate = add the batch dimension).
    NuScenesViDARDatasetV1 with the released pipeline (config :279-330, :332-375)."""

    def __init__(self, ann_file, data_root="", queue_length=4, future_length=1, test_mode=False,
                 load_interval=1, load_frame_interval=None, rand_frame_interval=(1,),
                 ego_mask=(-0.8, -1.5, 0.8, 2.5), sweeps_num=2, voxel_size=(1.0, 1.0, 1.0),
                 point_cloud_range=PC_RANGE, max_voxels=50000, dataset="nuscenes", augment=None, img_scale=None):
        self.infos, self.metadata = load_infos(ann_file, load_interval)
        self.data_root, self.dataset = str(data_root), dataset
        self.queue_length, self.future_length, self.test_mode = queue_length, future_length, test_mode
        self.rand_frame_interval, self.ego_mask = tuple(rand_frame_interval), ego_mask
        self.sweeps_num, self.voxel_size = sweeps_num, voxel_size
        self.point_cloud_range, self.max_voxels = point_cloud_range, max_voxels
        # True: the released training augmentation (nuScenes: photometric + crop/resize/flip, config :308-311;
        # OpenScene: photometric only, OpenScene config :297-299); a callable (imgs, meta, aug_param) -> imgs; None: none
        self.augment = TrainAugment(crop_resize_flip=dataset == "nuscenes") if augment is True else augment
        # OpenScene resizes the normalised images by 2/3 (RandomScaleImageMultiViewImage) in train AND test pipelines
        self.img_scale = (2.0 / 3.0 if dataset == "nuplan" else None) if img_scale is None else img_scale
        self.usable_index = usable_indices(self.infos, future_length, queue_length, test_mode, load_frame_interval)

    def __len__(self):
        return len(self.usable_index)

    def _path(self, p):
        return str(Path(self.data_root) / p) if self.data_root and not Path(p).is_absolute() else str(p)

    def frame(self, index, with_images=True, aug_param=None):
        """one frame through the pipeline -> record dict(img, points, img_metas[, aug_param])"""
        meta = frame_meta_from_info(copy.deepcopy(self.infos[index]), self.dataset, self.data_root)
        path = self._path(meta["pts_filename"])
        pts = load_pcd_file(path) if path.endswith(".pcd") else load_points_file(path)
        if not self.test_mode and self.dataset == "nuscenes":        # train pipeline: sweeps + voxel subsample
            sweeps = [dict(s, data_path=self._path(s["data_path"])) for s in meta.get("sweeps", [])]
            pts = load_multi_sweeps(pts, sweeps, meta["timestamp"], self.sweeps_num, ego_mask=self.ego_mask)
            pts = voxel_point_sampler(pts, self.voxel_size, self.point_cloud_range, self.max_voxels)
        elif not self.test_mode:                                     # OpenScene: sweeps_num = 0, six columns
            pts = np.array(pts, dtype=np.float32, copy=True)         # LoadNuPlanPointsFromMultiSweeps with no sweeps:
            pts[:, 4] = 0                                            # key-frame time slot, then hard_sweeps_timestamp
            pts[:, -1] = 0                                           # on the LAST column (nuplan_loading.py:283-288)
            pts = voxel_point_sampler(pts, self.voxel_size, self.point_cloud_range, self.max_voxels)
        rec = dict(points=torch.from_numpy(np.ascontiguousarray(pts)), img_metas=meta)
        if with_images:
            imgs = load_raw_images([self._path(p) for p in meta["img_filename"]])
            if self.augment is not None and not self.test_mode:
                aug_param = {} if aug_param is None else aug_param
                imgs = self.augment(imgs, meta, aug_param)
                rec["aug_param"] = aug_param
            img, shape = normalise_pad(imgs, scale=self.img_scale)
            if self.img_scale is not None:                           # transform_3d.py:317-323
                k = np.eye(4); k[0, 0] = k[1, 1] = self.img_scale
                meta["lidar2img"] = [k @ np.asarray(a) for a in meta["lidar2img"]]
            n = img.shape[0]
            meta.update(img_shape=[shape] * n, pad_shape=[shape] * n, img_norm_cfg=dict(IMG_NORM))
            rec["img"] = img
        return rec

    def _prepare(self, index, rand_interval=None):
        interval = int(np.random.choice(self.rand_frame_interval, 1)[0]) if rand_interval is None else rand_interval
        prev, fut = frame_index_lists(index, self.queue_length, self.future_length, interval, len(self.infos))
        previous_queue, aug_param = [], None
        for k in prev:                       # the first frame draws the augmentation, the others replay it (:116-124)
            rec = self.frame(k, aug_param=aug_param)
            aug_param = copy.deepcopy(rec["aug_param"]) if "aug_param" in rec else None
            previous_queue.append(rec)
        future_queue = [self.frame(k, with_images=False) for k in fut]
        return union2one(previous_queue, future_queue, self.future_length, self.ego_mask)

    def __getitem__(self, i):
        """the template's retry protocol (:199-219): a sample whose future leaves its scene is retried once with
        frame interval 1, then replaced by a random other sample (training) / the next one (testing)"""
        rand_interval = None
        while True:
            data = self._prepare(self.usable_index[i], rand_interval)
            if data is not None:
                return data
            if self.test_mode:
                i += 1
            elif rand_interval is None:
                rand_interval = 1
            else:
                i = int(np.random.choice(len(self.usable_index)))       # [3P] mmdet3d _rand_another (single group)
                rand_interval = None
