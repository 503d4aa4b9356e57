"""Generate tests/golden/dvr_family_*.npz from the REFERENCE's own kernels compiled for the host
(oracle/_ref, see oracle/build_ref.py).  Run in the build container (needs /root/reference):

    python tests/golden/make_dvr_golden.py

Stored compactly: per-ray live count + concatenated live prefixes of the padded rows."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import build_ref  # noqa: E402
from dvr_cases import case, compact, expand  # noqa: E402

ts = torch.from_numpy
build_ref.build(verbose=False)
dvr, dvxlr2 = build_ref.load("ref_dvr"), build_ref.load("ref_dvxlr_v2")
dvxlr = build_ref.load("ref_dvxlr")
for name in ["two_frames", "static_sigma", "small_grid"]:
    sigma, origin, points, tindex = case(name)
    regul = np.random.default_rng(7).standard_normal(sigma.shape).astype(np.float32)
    a = [t.numpy() for t in dvxlr2.render_v2(ts(sigma), ts(origin), ts(points), ts(tindex), ts(regul))]
    v1 = [t.numpy() for t in dvxlr.render(ts(sigma), ts(origin), ts(points), ts(tindex))]
    assert all(np.array_equal(x, y) for x, y in zip(v1, a[:4]))       # v2 == v1 on shared outputs
    cnt, dd_c, idx_c, rp_c, ind_c = compact(a[2], a[3], extra=(a[4], a[5]))
    back = expand(cnt, dd_c, idx_c, [(rp_c, 0.0), (ind_c, -1.0)])
    assert all(np.array_equal(x, y) for x, y in zip(back, a[2:]))     # compaction is lossless
    f = [t.numpy() for t in dvr.render_forward(ts(sigma), ts(origin), ts(points), ts(tindex),
                                               list(sigma.shape[1:]), "train")]
    r = [t.numpy() for t in dvr.render(ts(sigma), ts(origin), ts(points), ts(tindex), "l2")]
    out = Path(__file__).parent / f"dvr_family_{name}.npz"
    np.savez_compressed(out, pred=a[0], gt=a[1], count=cnt, dd=dd_c, idx=idx_c, ray_pred=rp_c,
                        indicator=ind_c, fwd_pred=f[0], fwd_gt=f[1], dvr_pred=r[0], dvr_gt=r[1],
                        dvr_grad_zsum=r[2].sum(axis=(2, 3)))
    print(out, out.stat().st_size, "bytes; rays", cnt.size, "mean count", cnt.mean())
