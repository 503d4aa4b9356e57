"""CPU: vidar_amd.data.assemble (usable-index scan, frame index lists, union2one) against a golden
produced by the reference's own dataset source text (tests/golden/make_union2one_golden.py)."""
import copy
import pickle
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))


@pytest.fixture(scope="module")
def gold():
    with open(GOLD / "union2one.pkl", "rb") as fh:
        return pickle.load(fh)


def test_usable_indices_match_reference(gold):
    from vidar_amd.data import usable_indices
    infos = [dict(scene_token=t) for t in gold["scenes"]]
    for (test_mode, q, f), want in gold["scans"].items():
        assert usable_indices(infos, f, q, test_mode) == want, (test_mode, q, f)
    assert usable_indices(infos, 2, 3, False, load_frame_interval=3) == gold["scans"][(False, 3, 2)][::3]


def test_frame_index_lists_match_reference(gold):
    from vidar_amd.data import frame_index_lists
    n = len(gold["scenes"])
    for (q, f, index, ri), (prev, fut) in gold["lists"].items():
        assert frame_index_lists(index, q, f, ri, n) == (prev, fut), (q, f, index, ri)


@pytest.mark.parametrize("name", ["plain", "ego_mask", "new_scene_in_history", "scene_ends"])
def test_union2one_matches_reference(gold, name):
    from make_union2one_golden import records
    from vidar_amd.data import union2one
    case = gold["cases"][name]
    seed, n, brk, q, f, mask = case["args"]
    recs = records(seed, n, brk)
    got = union2one(copy.deepcopy(recs[:q + 1]), copy.deepcopy(recs[q:q + 1 + f]), f, ego_mask=mask)
    want = case["ret"]
    if want is None:
        assert got is None
        return
    assert sorted(k for k in got if k != "img_metas") == want["keys"]
    np.testing.assert_array_equal(got["img"].numpy(), want["img"])
    np.testing.assert_array_equal(got["gt_points"].numpy(), want["gt_points"])
    assert got["gt_points"].dtype == torch.float32
    assert sorted(got["img_metas"]) == sorted(want["metas"])
    for i, wm in want["metas"].items():
        gm = got["img_metas"][i]
        assert sorted(gm) == sorted(wm), i
        for k, v in wm.items():
            if isinstance(v, np.ndarray) and v.dtype.kind in "fiu":
                np.testing.assert_allclose(np.asarray(gm[k], np.float64), v, rtol=0, atol=1e-12, err_msg=f"{i}.{k}")
            else:
                assert np.all(np.asarray(gm[k] == v)), (i, k)


def test_assembled_sample_drives_the_model(gold):
    """the assembled metas carry every field ViDAR.forward_train reads"""
    from make_union2one_golden import records
    from vidar_amd.data import union2one
    recs = records(1, 7, None)
    s = union2one(recs[:4], recs[3:6], 2)
    last = s["img_metas"][3]
    for k in ("future_can_bus", "future2ref_lidar_transform", "ref2future_lidar_transform",
              "total_cur2ref_lidar_transform", "total_ref2cur_lidar_transform", "ref_lidar_to_cur_lidar",
              "prev_bev_exists", "can_bus"):
        assert k in last
    assert last["future2ref_lidar_transform"].shape == (3, 4, 4)
    assert last["total_cur2ref_lidar_transform"].shape == (6, 4, 4)      # 3 history + (1 + 2) futures
    # frame 3 (the reference frame) maps onto itself
    np.testing.assert_allclose(last["total_cur2ref_lidar_transform"][3], np.eye(4), atol=1e-9)
    assert [m["prev_bev_exists"] for m in s["img_metas"].values()] == [False, True, True, True]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frame_meta_from_info_matches_reference(gold, seed):
    from make_union2one_golden import info_record
    from vidar_amd.data import frame_meta_from_info
    got = frame_meta_from_info(info_record(seed))
    want = gold["infos"][seed]
    assert sorted(got) == sorted(want)
    for k, w in want.items():
        if isinstance(w, (list, np.ndarray)) and len(w) and not isinstance(w[0], str):
            np.testing.assert_allclose(np.asarray(got[k], np.float64), np.asarray(w, np.float64), rtol=1e-12, atol=1e-12,
                                       err_msg=k)
        else:
            assert got[k] == w, k


@pytest.mark.parametrize("seed", [0, 1])
def test_frame_meta_from_nuplan_info_matches_reference(gold, seed):
    from make_union2one_golden import info_record
    from vidar_amd.data import frame_meta_from_info
    rec = info_record(seed)
    rec["sample_prev"], rec["sample_next"] = rec.pop("prev"), rec.pop("next")
    got = frame_meta_from_info(rec, dataset="nuplan", data_root="data/openscene")
    want = gold["infos_nuplan"][seed]
    assert sorted(got) == sorted(want)
    for k, w in want.items():
        if isinstance(w, (list, np.ndarray)) and len(w) and not isinstance(w[0], str):
            np.testing.assert_allclose(np.asarray(got[k], np.float64), np.asarray(w, np.float64), rtol=1e-12, atol=1e-12,
                                       err_msg=k)
        else:
            assert got[k] == w, k
    with pytest.raises(ValueError):
        frame_meta_from_info(info_record(0), dataset="kitti")
