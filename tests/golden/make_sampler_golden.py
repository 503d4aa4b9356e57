"""Golden index lists from the reference's own DistributedGroupSampler / DistributedSampler
(projects/mmdet3d_plugin/datasets/samplers/*.py) executed here (mmcv.runner.get_dist_info, the SAMPLER registry
and IPython stubbed).   python tests/golden/make_sampler_golden.py  ->  tests/golden/sampler.json"""
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
CASES = [(23, 1, 4, 0), (23, 1, 4, 7), (10, 2, 3, 1), (5, 1, 8, 0), (64, 1, 8, 3)]   # (n, samples_per_gpu, world, seed)


class DS:
    def __init__(self, n):
        self.flag = np.zeros(n, dtype=np.uint8)
        self.n = n

    def __len__(self):
        return self.n


def main():
    import ref_import as R
    R.install_stubs()
    sys.modules["mmcv.runner"].get_dist_info = lambda: (0, 1)
    sys.modules.setdefault("IPython", types.ModuleType("IPython")).embed = lambda *a, **k: None
    pkg = types.ModuleType("refsamplers"); pkg.__path__ = []
    sys.modules["refsamplers"] = pkg
    sm = types.ModuleType("refsamplers.sampler"); sm.SAMPLER = R._Registry(); sys.modules["refsamplers.sampler"] = sm
    mods = {}
    for name in ("group_sampler", "distributed_sampler"):
        spec = importlib.util.spec_from_file_location(f"refsamplers.{name}", R.PLUGIN / f"datasets/samplers/{name}.py")
        m = importlib.util.module_from_spec(spec); sys.modules[f"refsamplers.{name}"] = m; spec.loader.exec_module(m)
        mods[name] = m
    out = {}
    for n, spg, world, seed in CASES:
        for epoch in (0, 3):
            for rank in range(world):
                s = mods["group_sampler"].DistributedGroupSampler(DS(n), spg, world, rank, seed)
                s.set_epoch(epoch)
                out[f"train/{n}/{spg}/{world}/{seed}/{epoch}/{rank}"] = [int(i) for i in s]
        for rank in range(world):
            s = mods["distributed_sampler"].DistributedSampler(DS(n), num_replicas=world, rank=rank, shuffle=False)
            out[f"test/{n}/{world}/{rank}"] = [int(i) for i in s]
    (HERE / "sampler.json").write_text(json.dumps(out))


if __name__ == "__main__":
    main()
