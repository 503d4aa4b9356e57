"""Golden for vidar_amd.data.reader.load_pcd_file from the reference's own PointCloud / LoadNuPlanPointsFromFile
(projects/mmdet3d_plugin/datasets/pipelines/nuplan_loading.py:26-225) executed here on a synthetic binary .pcd
(mixed field types, a comment line, trailing garbage).   python tests/golden/make_pcd_golden.py -> tests/golden/pcd.npz"""
import importlib.util
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))


def write_pcd(path, seed=0, n=257):
    rng = np.random.default_rng(seed)
    row = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("intensity", np.uint8),
                    ("ring", np.uint8), ("lidar_info", np.uint8), ("extra", np.float64)])
    pts = np.zeros(n, row)
    for k in ("x", "y", "z"):
        pts[k] = rng.uniform(-60, 60, n)
    pts["intensity"] = rng.integers(0, 255, n); pts["ring"] = rng.integers(0, 40, n)
    pts["lidar_info"] = rng.integers(0, 5, n); pts["extra"] = rng.standard_normal(n)
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity ring lidar_info extra\n"
              "SIZE 4 4 4 1 1 1 8\nTYPE F F F U U U F\nCOUNT 1 1 1 1 1 1 1\n"
              f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(header.encode()); f.write(pts.tobytes()); f.write(b"\x00" * 37)
    return path


def main():
    import ref_import as R
    R.install_stubs()
    from make_loading_golden import Points
    sys.modules["mmcv"].FileClient = type("FileClient", (), {"__init__": lambda self, **k: None})
    R._mod("mmdet.datasets"); R._mod("mmdet.datasets.builder", PIPELINES=R._Registry())
    R._mod("mmdet3d.core"); R._mod("mmdet3d.core.points", BasePoints=Points,
                                   get_points_type=lambda t: (lambda p, points_dim=None, attribute_dims=None: Points(p)))
    R._mod("mmdet3d.datasets"); R._mod("mmdet3d.datasets.pipelines", VoxelBasedPointSampler=type("VoxelBasedPointSampler", (), {}))
    pkg = types.ModuleType("refpipes"); pkg.__path__ = []; sys.modules["refpipes"] = pkg
    mods = {}
    for name in ("loading", "nuplan_loading"):
        spec = importlib.util.spec_from_file_location(f"refpipes.{name}", R.PLUGIN / f"datasets/pipelines/{name}.py")
        m = importlib.util.module_from_spec(spec); sys.modules[f"refpipes.{name}"] = m; spec.loader.exec_module(m)
        mods[name] = m
    with tempfile.TemporaryDirectory() as d:
        f = write_pcd(Path(d) / "a.pcd")
        res = mods["nuplan_loading"].LoadNuPlanPointsFromFile("LIDAR")(dict(pts_filename=str(f)))
        pts = res["points"].tensor.numpy()
        loader = mods["nuplan_loading"].LoadNuPlanPointsFromMultiSweeps(
            sweeps_num=0, use_dim=[0, 1, 2, 3, 4, 5], file_client_args=dict(backend="disk"), pad_empty_sweeps=True,
            remove_close=True, ego_mask=(-0.8, -1.5, 0.8, 2.5), hard_sweeps_timestamp=0, random_select=False)
        train = loader(dict(points=Points(pts), sweeps=[], timestamp=1.0))["points"].tensor.numpy()
    np.savez_compressed(HERE / "pcd.npz", points=pts, train_points=train)


if __name__ == "__main__":
    main()
