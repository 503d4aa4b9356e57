"""Golden for vidar_amd.evaluate.summarize: the reference's own NuScenesViDARDatasetTemplate.evaluate
(projects/mmdet3d_plugin/datasets/nuscenes_vidar_dataset_template.py:147-196) run on seeded
per-sample result dicts.  Run in the container that has /root/reference:  python tests/golden/make_evaluate_golden.py"""
import copy
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_import as R  # noqa: E402


def reference_evaluate():
    R.install_stubs()
    import mmcv
    mmcv.track_iter_progress = lambda x: x
    # the template imports mmdet3d / nuscenes symbols at module level: stub what the import touches
    for name, attrs in {
        "mmdet.datasets": dict(DATASETS=R._Registry()),
        "mmdet3d.datasets": dict(NuScenesDataset=object),
        "mmcv.parallel": dict(DataContainer=object),
        "nuscenes": {}, "nuscenes.eval": {}, "nuscenes.eval.common": {},
        "nuscenes.eval.common.utils": dict(quaternion_yaw=None, Quaternion=None),
    }.items():
        if name not in sys.modules:
            R._mod(name, **attrs)
    src = (R.PLUGIN / "datasets/nuscenes_vidar_dataset_template.py").read_text()
    # keep only the evaluate method: exec the def inside a throw-away class
    start = src.index("    def evaluate(self,")
    end = src.index("    def __getitem__(self, idx):")
    ns = {"mmcv": mmcv}
    exec("class T:\n" + src[start:end], ns)
    return ns["T"]().evaluate


def main():
    rng = np.random.default_rng(11)
    results = []
    for s in range(7):
        res = {}
        for f in range(4):
            c = int(rng.integers(1, 3))
            res[f"frame.{f}"] = dict(count=c, chamfer_distance=float(rng.uniform(0, 5)) * c,
                                     l1_error=float(rng.uniform(0, 9)) * c,
                                     absrel_error=float(rng.uniform(0, 1)) * c)
        results.append(res)
    expected = reference_evaluate()(copy.deepcopy(results))
    (HERE / "evaluate_summary.json").write_text(json.dumps(dict(results=results, expected=expected), indent=1))
    print("wrote evaluate_summary.json")


if __name__ == "__main__":
    main()
