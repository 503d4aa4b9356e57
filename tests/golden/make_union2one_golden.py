"""Golden for vidar_amd.data.assemble from the reference's own dataset code: the source text of
NuScenesViDARDatasetV1.union2one/_mask_points (datasets/nuscenes_vidar_dataset_v1.py:22-203) and of
the template's usable-index scan / frame index lists (nuscenes_vidar_dataset_template.py:44-68,
:101-135) executed in throw-away classes.  nuscenes-devkit / pyquaternion / mmcv DataContainer are
not installable: `transform_matrix`, `Quaternion` and `DC` are restated here ([3P]).
    python tests/golden/make_union2one_golden.py"""
import copy
import pickle
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/projects/mmdet3d_plugin/datasets")


class Quaternion:                       # [3P] pyquaternion, (w, x, y, z)
    def __init__(self, q):
        self.q = np.asarray(q, np.float64)

    @property
    def rotation_matrix(self):
        w, x, y, z = self.q / np.linalg.norm(self.q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def transform_matrix(translation=np.array([0, 0, 0]), rotation=Quaternion([1, 0, 0, 0]), inverse=False):
    """[3P] nuscenes.utils.geometry_utils.transform_matrix"""
    tm = np.eye(4)
    if inverse:
        rot_inv = rotation.rotation_matrix.T
        trans = np.transpose(-np.array(translation))
        tm[:3, :3] = rot_inv
        tm[:3, 3] = rot_inv.dot(trans)
    else:
        tm[:3, :3] = rotation.rotation_matrix
        tm[:3, 3] = np.transpose(np.array(translation))
    return tm


def quaternion_yaw(q):                  # [3P] nuscenes.eval.common.utils
    v = np.dot(q.rotation_matrix, np.array([1, 0, 0]))
    return np.arctan2(v[1], v[0])


def _quat_array(self, dtype=None, copy=None):
    return np.asarray(self.q, dtype)
Quaternion.__array__ = _quat_array      # pyquaternion converts to (w, x, y, z) when assigned into an array
Quaternion.__len__ = lambda self: 4
Quaternion.__getitem__ = lambda self, i: self.q[i]


def info_record(seed):
    rng = np.random.default_rng(seed)
    def quat():
        q = rng.normal(size=4); return (q / np.linalg.norm(q)).tolist()
    cams = {}
    for c in ("CAM_FRONT", "CAM_BACK"):
        a = rng.normal(size=(3, 3)); r, _ = np.linalg.qr(a)
        cams[c] = dict(data_path=f"{c}.jpg", sensor2lidar_rotation=r, sensor2lidar_translation=rng.normal(size=3),
                       cam_intrinsic=np.array([[1266.4, 0, 816.2], [0, 1266.4, 491.5], [0, 0, 1]]))
    return dict(token=f"tok{seed}", lidar_path="x.bin", sweeps=[], ego2global_translation=rng.normal(size=3).tolist(),
                ego2global_rotation=quat(), lidar2ego_translation=[0.94, 0.0, 1.84], lidar2ego_rotation=quat(),
                prev="p", next="n", scene_token="sc", can_bus=rng.normal(size=18), frame_idx=3,
                timestamp=1533151603547590, cams=cams)


class DC:                               # [3P] mmcv.parallel.DataContainer, as far as union2one uses it
    def __init__(self, data, cpu_only=False, stack=False):
        self.data = data


def records(seed, n, scene_break_at=None):
    """n consecutive frame records with smooth ego motion; optional scene change."""
    rng = np.random.default_rng(seed)
    out = []
    x = y = yaw = 0.0
    for k in range(n):
        x += rng.normal(2.0, 0.5); y += rng.normal(0.0, 0.5); yaw += np.deg2rad(rng.normal(0, 3.0))
        q = [np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)]
        can_bus = rng.normal(size=18)
        can_bus[:3] = [x, y, 0.0]
        can_bus[-1] = np.rad2deg(yaw)
        pts = rng.uniform(-20, 20, (30 + k, 5)).astype(np.float32)
        pts[:3, :2] = rng.uniform(-0.5, 0.5, (3, 2))          # a few points on the ego vehicle
        meta = dict(scene_token="A" if scene_break_at is None or k < scene_break_at else "B", can_bus=can_bus,
                    ego2global_translation=[x, y, 0.1 * k], ego2global_rotation=q,
                    lidar2ego_translation=[0.9, 0.0, 1.8], lidar2ego_rotation=[0.7071067811865476, 0, 0, 0.7071067811865476],
                    sample_idx=f"s{seed}_{k}")
        out.append(dict(img=torch.full((2, 3, 4, 6), float(k)), points=torch.from_numpy(pts), img_metas=meta,
                        aug_param=dict(k=k)))
    return out


def wrap(rec):
    r = copy.deepcopy(rec)
    return dict(img=DC(r["img"]), points=DC(r["points"]), img_metas=DC(r["img_metas"]), aug_param=r["aug_param"])


def main():
    src = (REF / "nuscenes_vidar_dataset_v1.py").read_text()
    a = src.index("    def _mask_points(self, pts_list):")
    ns = dict(np=np, torch=torch, copy=copy, transform_matrix=transform_matrix, Quaternion=Quaternion, DC=DC)
    exec("class V1:\n" + src[a:], ns)
    tsrc = (REF / "nuscenes_vidar_dataset_template.py").read_text()

    cases = {}
    for name, (seed, n, brk, q, f, mask) in dict(
            plain=(1, 7, None, 3, 2, None), ego_mask=(2, 7, None, 3, 2, [-0.8, -1.5, 0.8, 2.5]),
            new_scene_in_history=(3, 7, 2, 3, 2, None), scene_ends=(4, 7, 5, 3, 2, None)).items():
        recs = records(seed, n, brk)
        prev, fut = recs[:q + 1], recs[q:q + 1 + f]
        ds = ns["V1"]()
        ds.ego_mask, ds.future_length = mask, f
        ret = ds.union2one([wrap(r) for r in prev], [wrap(r) for r in fut])
        if ret is None:
            cases[name] = dict(args=(seed, n, brk, q, f, mask), ret=None)
            continue
        metas = ret["img_metas"].data
        cases[name] = dict(args=(seed, n, brk, q, f, mask),
                           ret=dict(img=ret["img"].data.numpy(), gt_points=ret["gt_points"].data.numpy(),
                                    keys=sorted(k for k in ret if k != "img_metas"),
                                    metas={i: {k: (np.asarray(v) if not isinstance(v, (str, bool, dict, type(None))) else v)
                                               for k, v in m.items()} for i, m in metas.items()}))

    # usable-index scan + frame index lists
    a = tsrc.index("        last_scene_index = None")
    b = tsrc.index("        # Remove useless frame index if load_frame_interval is assigned.")
    scan = "def scan(self):\n" + tsrc[a:b].replace("mmcv.track_iter_progress(self.data_infos)", "self.data_infos") + "        return usable_index\n"
    exec(scan, ns)
    a = tsrc.index("        previous_index_list = list(range(")
    b = tsrc.index("        aug_param = None\n        for i in previous_index_list:")
    c = tsrc.index("        future_index_list = list(range(")
    d = tsrc.index("        has_future = False")
    idx = ("def lists(self, index, rand_interval):\n" + tsrc[a:b] + tsrc[c:d] +
           "        cl = lambda i: min(max(0, i), len(self.data_infos) - 1)\n"
           "        return [cl(i) for i in previous_index_list], [cl(i) for i in future_index_list]\n")
    exec(idx, ns)
    infos = [dict(scene_token=t) for t in "AAAAAABBBBCCCCCCCCDD"]
    scans, lists = {}, {}
    for test_mode in (False, True):
        for q, f in ((3, 2), (4, 6), (1, 1)):
            o = type("O", (), {})()
            o.data_infos, o.queue_length, o.future_length, o.test_mode = infos, q, f, test_mode
            scans[(test_mode, q, f)] = ns["scan"](o)
            for index in (0, 5, 12, 19):
                for ri in (1, 2, -1):
                    lists[(q, f, index, ri)] = ns["lists"](o, index, ri)
    # get_data_info (nuscenes_dataset.py:153-227) + the template's extra fields
    dsrc = (REF / "nuscenes_dataset.py").read_text()
    a = dsrc.index("        info = self.data_infos[index]\n        # standard protocal modified from SECOND.Pytorch")
    b = dsrc.index("        return input_dict\n\n    def __getitem__(self, idx):")
    ns["quaternion_yaw"] = quaternion_yaw
    exec("def get_data_info(self, index):\n" + dsrc[a:b] + "        return input_dict\n", ns)
    infos_out = {}
    for seed in (0, 1, 2):
        o = type("O", (), {})()
        o.data_infos, o.modality, o.test_mode, o.include_test = [info_record(seed)], dict(use_camera=True), True, False
        d = ns["get_data_info"](o, 0)
        d.update(lidar2ego_translation=o.data_infos[0]["lidar2ego_translation"],
                 lidar2ego_rotation=o.data_infos[0]["lidar2ego_rotation"], cam2img=d["cam_intrinsic"])
        infos_out[seed] = d
    # nuPlan / OpenScene flavour (nuplan_vidar_dataset_template.py:48-118)
    psrc = (REF / "nuplan_vidar_dataset_template.py").read_text()
    a = psrc.index("        info = self.data_infos[index]\n        # standard protocal modified from SECOND.Pytorch")
    b = psrc.index("        can_bus[-1] = patch_angle") + len("        can_bus[-1] = patch_angle")
    import os
    ns["os"] = os
    exec("def get_data_info_nuplan(self, index):\n" + psrc[a:b] + "\n        return input_dict\n", ns)
    infos_nuplan = {}
    for seed in (0, 1):
        o = type("O", (), {})()
        rec = info_record(seed)
        rec["sample_prev"], rec["sample_next"] = rec.pop("prev"), rec.pop("next")
        o.data_infos, o.data_root = [rec], "data/openscene"
        infos_nuplan[seed] = ns["get_data_info_nuplan"](o, 0)
    with open(HERE / "union2one.pkl", "wb") as fh:
        pickle.dump(dict(infos=infos_out, infos_nuplan=infos_nuplan, cases=cases, scenes="AAAAAABBBBCCCCCCCCDD", scans=scans, lists=lists), fh, protocol=4)
    print("wrote union2one.pkl", {k: (None if v["ret"] is None else v["ret"]["gt_points"].shape) for k, v in cases.items()},
          len(scans), len(lists))


if __name__ == "__main__":
    main()
