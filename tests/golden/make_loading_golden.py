"""Golden for vidar_amd.data.reader.load_multi_sweeps from the reference's own
CustomLoadPointsFromMultiSweeps (projects/mmdet3d_plugin/datasets/pipelines/loading.py:10-223) executed in
this container on synthetic sweep files.  mmcv / mmdet / mmdet3d are absent: the few symbols the file
touches are stood in for ([3P]: FileClient -> disk bytes, BasePoints -> a minimal tensor wrapper with the
operations the loader uses: .tensor, boolean / column indexing, new_point, cat).
    python tests/golden/make_loading_golden.py  ->  tests/golden/loading.npz"""
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))


class Points:                                   # [3P] stand-in for mmdet3d.core.points.LiDARPoints
    def __init__(self, tensor):
        self.tensor = torch.as_tensor(np.asarray(tensor), dtype=torch.float32).clone()

    def __getitem__(self, item):
        if isinstance(item, tuple):
            rows, cols = item
            return Points(self.tensor[rows][:, cols])
        if isinstance(item, np.ndarray):
            item = torch.from_numpy(item)
        return Points(self.tensor[item])

    def new_point(self, data):
        return Points(data)

    def cat(self, pts):
        return Points(torch.cat([p.tensor for p in pts], 0))


def synthetic_case(root, seed, n_sweeps):
    rng = np.random.default_rng(seed)
    key = rng.uniform(-30, 30, (400, 5)).astype(np.float32)
    key[:40, :2] = rng.uniform(-0.9, 0.9, (40, 2))              # inside the close radius
    key[40:80, 0] = rng.uniform(-0.8, 0.8, 40); key[40:80, 1] = rng.uniform(-1.5, 2.5, 40)   # ego box
    ts = 1000.5 + seed
    sweeps = []
    for k in range(n_sweeps):
        p = rng.uniform(-30, 30, (300, 5)).astype(np.float32)
        p[:30, :2] = rng.uniform(-0.95, 0.95, (30, 2))
        p[30:60, 0] = rng.uniform(-0.8, 0.8, 30); p[30:60, 1] = rng.uniform(1.0, 2.5, 30)
        f = Path(root) / f"s{seed}_{k}.bin"
        p.tofile(f)
        a = rng.uniform(-0.2, 0.2)
        rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        sweeps.append(dict(data_path=str(f), timestamp=(ts - 0.05 * (k + 1) * (1 + 0.1 * rng.random())) * 1e6,
                           sensor2lidar_rotation=rot, sensor2lidar_translation=rng.uniform(-1, 1, 3)))
    return key, sweeps, ts


CASES = [(0, 5), (1, 2), (2, 0), (3, 1)]          # (seed, number of sweeps available)
KW = dict(sweeps_num=2, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True, remove_close=True,
          ego_mask=(-0.8, -1.5, 0.8, 2.5), hard_sweeps_timestamp=0, random_select=False)


def main():
    import ref_import as R
    R.install_stubs()

    class FileClient:
        def __init__(self, **kw): pass
        def get(self, path): return Path(path).read_bytes()
    sys.modules["mmcv"].FileClient = FileClient
    sys.modules["mmcv"].check_file_exist = lambda p: None
    R._mod("mmdet.datasets"); R._mod("mmdet.datasets.builder", PIPELINES=R._Registry())
    R._mod("mmdet3d.core"); R._mod("mmdet3d.core.points", BasePoints=Points, get_points_type=lambda t: Points)
    R._mod("mmdet3d.datasets"); R._mod("mmdet3d.datasets.pipelines", VoxelBasedPointSampler=type("VoxelBasedPointSampler", (), {}))
    L = R.load_file("ref_loading", R.PLUGIN / "datasets/pipelines/loading.py")
    out = {}
    with tempfile.TemporaryDirectory() as root:
        for seed, n in CASES:
            key, sweeps, ts = synthetic_case(root, seed, n)
            loader = L.CustomLoadPointsFromMultiSweeps(file_client_args=dict(backend="disk"), **KW)
            res = loader(dict(points=Points(key), sweeps=sweeps, timestamp=ts))
            out[f"points{seed}"] = res["points"].tensor.numpy()
    np.savez_compressed(HERE / "loading.npz", **out)


if __name__ == "__main__":
    main()
