"""CPU: the info-pkl reader (vidar_amd/data/reader.py) -- multi-sweep loading pinned against the reference's
own CustomLoadPointsFromMultiSweeps text (tests/golden/make_loading_golden.py), voxel subsampling / image
normalisation against their definitions, and a synthetic mini nuScenes on disk read end to end into the
forward_train kwargs (through the same union2one that tests/test_assemble_cpu.py pins to the reference)."""
import pickle
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))


@pytest.mark.parametrize("seed,n", [(0, 5), (1, 2), (2, 0), (3, 1)])
def test_multi_sweep_loader_matches_reference_text(tmp_path, seed, n):
    from make_loading_golden import synthetic_case
    from vidar_amd.data.reader import load_multi_sweeps
    gold = np.load(GOLD / "loading.npz")
    key, sweeps, ts = synthetic_case(tmp_path, seed, n)
    got = load_multi_sweeps(key, sweeps, ts, sweeps_num=2, ego_mask=(-0.8, -1.5, 0.8, 2.5),
                            hard_sweeps_timestamp=0, random_select=False)
    np.testing.assert_array_equal(got, gold[f"points{seed}"])


def test_voxel_subsample_keeps_first_point_per_voxel_in_order():
    from vidar_amd.data.reader import voxel_subsample
    pts = np.array([[0.2, 0.2, 0.1, 1, 0], [0.3, 0.3, 0.3, 2, 0],      # same 1 m voxel -> first stays
                    [60.0, 0.0, 0.0, 3, 0],                            # outside the range
                    [-51.2, -51.2, -5.0, 4, 0],                        # on the lower corner: inside
                    [51.2, 0.0, 0.0, 5, 0],                            # on the upper bound: outside
                    [1.5, 0.2, 0.1, 6, 0], [0.1, 0.1, 0.9, 7, 0]], np.float32)
    out = voxel_subsample(pts)
    np.testing.assert_array_equal(out[:, 3], [1, 4, 6])
    assert voxel_subsample(pts, max_voxels=2).shape[0] == 2
    # 0.5 m voxels (vidar_full configs): the first two points now fall into different voxels
    # (grid = round(102.4 / 0.5) = 205 cells: x = 51.2 lands in cell 204, inside)
    np.testing.assert_array_equal(voxel_subsample(pts, voxel_size=(0.5, 0.5, 0.5))[:, 3], [1, 2, 4, 5, 6, 7])


def test_image_loading_normalises_bgr_and_pads(tmp_path):
    from PIL import Image
    from vidar_amd.data.reader import load_images
    rgb = np.random.default_rng(0).integers(0, 255, (45, 70, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.png")
    img, shape = load_images([tmp_path / "a.png", tmp_path / "a.png"])
    assert img.shape == (2, 3, 64, 96) and shape == (64, 96, 3)
    bgr = rgb[..., ::-1].astype(np.float32) - np.array([103.530, 116.280, 123.675], np.float32)
    np.testing.assert_allclose(img[0, :, :45, :70].numpy(), bgr.transpose(2, 0, 1), rtol=0, atol=1e-4)
    assert float(img[0, :, 45:].abs().max()) == 0 and float(img[0, :, :, 70:].abs().max()) == 0


def _mini_nuscenes(root, n_frames=9, cams=2):
    """two scenes (6 + 3 frames) with lidar .bin files, sweeps and tiny camera images"""
    rng = np.random.default_rng(5)
    from PIL import Image
    infos = []
    for k in range(n_frames):
        scene = "scene-a" if k < 6 else "scene-b"
        lidar = root / f"lidar_{k}.bin"
        rng.uniform(-40, 40, (500, 5)).astype(np.float32).tofile(lidar)
        sw = root / f"sweep_{k}.bin"
        rng.uniform(-40, 40, (300, 5)).astype(np.float32).tofile(sw)
        cam_infos = {}
        for c in range(cams):
            p = root / f"img_{k}_{c}.png"
            Image.fromarray(rng.integers(0, 255, (40, 64, 3), dtype=np.uint8)).save(p)
            yaw = c * np.pi
            cam_infos[f"CAM_{c}"] = dict(
                data_path=str(p), cam_intrinsic=np.array([[50.0, 0, 32], [0, 50.0, 20], [0, 0, 1]]),
                sensor2lidar_rotation=np.array([[np.cos(yaw), 0, np.sin(yaw)], [np.sin(yaw), 0, -np.cos(yaw)], [0, -1.0, 0]]) @ np.eye(3),
                sensor2lidar_translation=np.array([0.5 * c, 0.0, 1.5]))
        a = 0.05 * k
        infos.append(dict(token=f"tok{k}", lidar_path=str(lidar), timestamp=int((100 + 0.5 * k) * 1e6),
                          sweeps=[dict(data_path=str(sw), timestamp=int((100 + 0.5 * k - 0.05) * 1e6),
                                       sensor2lidar_rotation=np.eye(3), sensor2lidar_translation=np.zeros(3))],
                          ego2global_translation=[2.0 * k, 0.3 * k, 0.0],
                          ego2global_rotation=[np.cos(a / 2), 0.0, 0.0, np.sin(a / 2)],
                          lidar2ego_translation=[0.9, 0.0, 1.8], lidar2ego_rotation=[1.0, 0.0, 0.0, 0.0],
                          prev="" if k in (0, 6) else f"tok{k - 1}", next="" if k in (5, n_frames - 1) else f"tok{k + 1}",
                          scene_token=scene, can_bus=np.zeros(18), frame_idx=k if k < 6 else k - 6, cams=cam_infos))
    order = rng.permutation(n_frames)                       # the reader must sort by timestamp
    with open(root / "infos.pkl", "wb") as f:
        pickle.dump(dict(infos=[infos[i] for i in order], metadata=dict(version="v1.0-mini")), f)
    return root / "infos.pkl"


def test_mini_dataset_end_to_end(tmp_path):
    from vidar_amd.data.reader import ViDARSequenceDataset
    ann = _mini_nuscenes(tmp_path)
    ds = ViDARSequenceDataset(ann, queue_length=2, future_length=1, test_mode=False)
    assert [i["token"] for i in ds.infos] == [f"tok{k}" for k in range(9)]
    # train mode: every frame with 1 future frame in its own scene (template :44-68)
    assert ds.usable_index == [0, 1, 2, 3, 4, 6, 7]
    np.random.seed(0)
    s = ds[3]                                               # index 3: history frames 1, 2 + current 3, future 4
    assert s["img"].shape == (3, 2, 3, 64, 64)
    assert sorted(s["img_metas"]) == [0, 1, 2]
    m = s["img_metas"][2]
    assert m["sample_idx"] == "tok3" and m["img_shape"] == [(64, 64, 3)] * 2
    assert [s["img_metas"][t]["prev_bev_exists"] for t in range(3)] == [False, True, True]
    assert m["future2ref_lidar_transform"].shape == (2, 4, 4) and len(m["lidar2img"]) == 2
    gt = s["gt_points"].numpy()
    assert gt.shape[1] == 5 and sorted(np.unique(gt[:, 4])) == [0, 1, 2, 3]     # 2 history + current + 1 future
    assert gt.shape[0] < 4 * 800                            # sweeps merged, then one point per 1 m voxel
    ego = (np.abs(gt[:, 0]) <= 0.8) & (gt[:, 1] >= -1.5) & (gt[:, 1] <= 2.5)
    assert not ego.any()
    # the scene boundary: a sample whose future would cross into scene-b does not exist; test mode needs history
    dt = ViDARSequenceDataset(ann, queue_length=2, future_length=1, test_mode=True)
    assert dt.usable_index == [2, 3, 4]
    t = dt[0]
    assert t["gt_points"].shape[1] == 5 and t["img"].shape[0] == 3


def test_model_consumes_a_read_sample(tmp_path):
    """reader -> collate -> ViDAR.forward_train kwargs (ops on the CPU oracle, tiny BEV)"""
    from oracle import cpu_ops
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    from vidar_amd.data.reader import ViDARSequenceDataset
    from vidar_amd.synthetic import fpn_features
    ann = _mini_nuscenes(tmp_path)
    ds = ViDARSequenceDataset(ann, queue_length=4, future_length=1)
    np.random.seed(0); torch.manual_seed(0)
    s = ds[ds.usable_index.index(4)]
    cfg = get_config("vidar_1_8_nusc_1future", bev_h=12, bev_w=12)
    cfg["model"]["pts_bbox_head"]["transformer"]["num_cams"] = 2
    cfg["model"]["pts_bbox_head"]["transformer"]["encoder"]["transformerlayers"]["attn_cfgs"][1]["num_cams"] = 2
    model = T.build_model(cfg).train()
    feats = fpn_features(0, 5, num_cams=2, shapes=[(8, 8), (4, 4), (2, 2), (1, 1)])
    with cpu_ops.patched():
        losses = model(return_loss=True, img_metas=[s["img_metas"]], gt_points=[s["gt_points"]], img_feats=feats)
    assert all(torch.isfinite(v) for v in losses.values()) and len(losses) == 10


@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_crop_resize_flip_matches_reference_class(seed):
    """CropResizeFlipImage against the reference's own class (tests/golden/make_augment_golden.py): same draws,
    the second frame of a queue replays the first frame's parameters, cam2img / lidar2img follow the image"""
    import copy
    import random
    from make_augment_golden import CONF, inputs
    from vidar_amd.data.augment import CropResizeFlipImage
    gold = np.load(GOLD / "augment.npz")
    random.seed(seed); np.random.seed(seed)
    aug = CropResizeFlipImage(CONF, training=True)
    aug_param = {}
    for frame in range(2):
        imgs, cam2img, lidar2cam = inputs(seed * 10 + frame)
        meta = dict(cam2img=copy.deepcopy(cam2img), lidar2cam=copy.deepcopy(lidar2cam))
        out = aug([i.copy() for i in imgs], meta, aug_param)
        np.testing.assert_array_equal(np.stack(out), gold[f"s{seed}_f{frame}_img"])
        np.testing.assert_allclose(np.stack(meta["cam2img"]), gold[f"s{seed}_f{frame}_cam2img"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.stack(meta["lidar2img"]), gold[f"s{seed}_f{frame}_lidar2img"], rtol=0, atol=1e-12)
    p = aug_param["CropResizeFlipImage_param"]
    np.testing.assert_allclose([p[0], p[1][0], p[1][1], float(p[3])], gold[f"s{seed}_param"])


def test_hsv_round_trip_and_photometric_draw_order():
    """[3P, unpinned] the restated BGR<->HSV pair is an exact inverse on float images; the distortion consumes numpy's
    generator in the reference's order (8 binary draws + their uniforms) and leaves shape / dtype alone"""
    from vidar_amd.data.augment import PhotoMetricDistortionMultiViewImage, bgr2hsv, hsv2bgr
    img = np.random.default_rng(0).uniform(0, 255, (20, 30, 3)).astype(np.float32)
    np.testing.assert_allclose(hsv2bgr(bgr2hsv(img)), img, rtol=0, atol=2e-3)
    hsv = bgr2hsv(np.array([[[0.0, 0.0, 255.0], [255.0, 0.0, 0.0], [10.0, 10.0, 10.0]]], np.float32))   # red, blue, grey
    np.testing.assert_allclose(hsv[0, :, 0], [0.0, 240.0, 0.0], atol=1e-4)
    np.testing.assert_allclose(hsv[0, :, 1], [1.0, 1.0, 0.0], atol=1e-6)
    np.random.seed(3)
    out = PhotoMetricDistortionMultiViewImage()([img.copy(), img.copy()])
    assert len(out) == 2 and out[0].shape == img.shape and out[0].dtype == np.float32
    assert not np.allclose(out[0], out[1])                 # every image draws its own distortion (the loop is per image)


def test_training_reader_applies_one_augmentation_to_the_whole_queue(tmp_path):
    import random
    from vidar_amd.data.reader import TrainAugment, ViDARSequenceDataset
    ann = _mini_nuscenes(tmp_path)
    conf = {"reisze": [24, 32], "crop": (0, 0, 64, 40), "H": 40, "W": 64, "rand_flip": True}
    ds = ViDARSequenceDataset(ann, queue_length=2, future_length=1, augment=TrainAugment(conf, photometric=False))
    random.seed(1); np.random.seed(1)
    s = ds[3]
    params = [s["img_metas"][t]["aug_param"]["CropResizeFlipImage_param"] for t in range(3)]
    assert params[0] == params[1] == params[2]
    resize, dims = params[0][0], params[0][1]
    assert s["img"].shape[-2:] == ((dims[1] + 31) // 32 * 32, (dims[0] + 31) // 32 * 32)
    k = s["img_metas"][2]["cam2img"][0]
    np.testing.assert_allclose(k[0, 0], 50.0 * resize)     # focal length follows the resize
    np.testing.assert_allclose(s["img_metas"][2]["lidar2img"][0], k @ s["img_metas"][2]["lidar2cam"][0])


def test_pcd_reader_matches_reference_class(tmp_path):
    """binary .pcd parsing (OpenScene / nuPlan lidar) against the reference's own PointCloud + LoadNuPlanPointsFromFile,
    and the key-frame handling of its multi-sweep loader with sweeps_num = 0 (tests/golden/make_pcd_golden.py)"""
    from make_pcd_golden import write_pcd
    from vidar_amd.data.reader import load_pcd_file
    gold = np.load(GOLD / "pcd.npz")
    pts = load_pcd_file(write_pcd(tmp_path / "a.pcd"))
    assert pts.dtype == np.float32 and pts.shape == (257, 6)
    np.testing.assert_array_equal(pts, gold["points"])
    train = pts.copy(); train[:, 4] = 0; train[:, -1] = 0            # what ViDARSequenceDataset.frame does for nuplan
    np.testing.assert_array_equal(train, gold["train_points"])
    bad = tmp_path / "ascii.pcd"
    bad.write_text("FIELDS x\nSIZE 4\nTYPE F\nCOUNT 1\nWIDTH 1\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 1\nDATA ascii\n1.0\n")
    with pytest.raises(RuntimeError):
        load_pcd_file(bad)


def test_openscene_image_scale_follows_into_lidar2img():
    from vidar_amd.data.reader import normalise_pad
    img = np.random.default_rng(0).uniform(0, 255, (90, 120, 3)).astype(np.float32)
    out, shape = normalise_pad([img], scale=2.0 / 3.0)
    assert out.shape == (1, 3, 64, 96) and shape == (64, 96, 3)       # 60 x 80 resized, padded to /32
    assert float(out[0, :, 60:].abs().max()) == 0 and float(out[0, :, :, 80:].abs().max()) == 0
    flat = np.full((90, 120, 3), 100.0, np.float32)
    out, _ = normalise_pad([flat], scale=2.0 / 3.0)
    np.testing.assert_allclose(out[0, :, :60, :80].numpy(),
                               np.broadcast_to((100.0 - np.array([103.530, 116.280, 123.675], np.float32))[:, None, None], (3, 60, 80)),
                               rtol=0, atol=1e-4)


def test_getitem_retries_when_the_future_leaves_the_scene(tmp_path):
    """rand_frame_interval = 2 makes the future of index 3 (frames 3, 5) fine but of index 4 (4, 6 -> scene-b) not:
    the reader retries with interval 1 like the template (:199-219) instead of returning None"""
    from vidar_amd.data.reader import ViDARSequenceDataset
    ds = ViDARSequenceDataset(_mini_nuscenes(tmp_path), queue_length=1, future_length=1, rand_frame_interval=(2,))
    np.random.seed(0)
    assert ds._prepare(4) is None and ds._prepare(4, rand_interval=1) is not None
    s = ds[ds.usable_index.index(4)]
    assert s is not None and s["img_metas"][1]["sample_idx"] == "tok4"


def test_voxel_point_sampler_shuffles_like_mmdet3d_before_sampling():
    """CustomVoxelBasedPointSampler.__call__ ([3P] mmdet3d VoxelBasedPointSampler, restated): the points are
    shuffled (twice when every point carries time 0: the empty previous-sweep set aliases the current array) before
    the first point per voxel is kept -- so which point survives is random, key-frame points do not always win,
    and the numpy stream advances exactly like two np.random.shuffle calls on the full array."""
    from vidar_amd.data.reader import voxel_point_sampler, voxel_subsample
    rng = np.random.default_rng(0)
    pts = np.zeros((400, 5), np.float32)
    pts[:, :3] = rng.uniform(-4, 4, (400, 3))
    pts[:, 3] = np.arange(400)                           # identity of a point
    np.random.seed(5)
    got = voxel_point_sampler(pts.copy())
    after = np.random.random()
    np.random.seed(5)
    ref = pts.copy()
    np.random.shuffle(ref); np.random.shuffle(ref)
    want = voxel_subsample(ref)
    np.testing.assert_array_equal(got, want)
    assert after == np.random.random(), "the numpy stream must advance by exactly two shuffles"
    first = voxel_subsample(pts)                         # deterministic order: first point of each voxel
    assert got.shape == first.shape and not np.array_equal(np.sort(got[:, 3]), np.sort(first[:, 3]))
    # points of earlier sweeps (time != 0) are dropped when no prev_sweep_cfg is given, like the released configs
    mixed = pts.copy(); mixed[200:, 4] = 0.05
    np.random.seed(6)
    out = voxel_point_sampler(mixed)
    assert (out[:, 3] < 200).all()


def test_named_recipes_carry_their_dataset_settings():
    """tools/train.py --ann-file must train the recipe it names: temporal augmentation, subset stride, GT voxel
    size and future length of the released configs (vidar_1_8_nusc_1future.py:14-24, :294; ..._3future.py:14-28,
    :301; vidar_full_nusc_1future.py:14-24; OpenScene mini: :14-28, :292), and a released config file overrides."""
    from vidar_amd.configs import dataset_kwargs, get_config
    k = dataset_kwargs(get_config("vidar_1_8_nusc_1future"))
    assert k["rand_frame_interval"] == (-1, 1) and k["load_frame_interval"] == 8 and k["voxel_size"] == (0.5,) * 3
    assert k["future_length"] == 2 and k["queue_length"] == 4 and k["dataset"] == "nuscenes" and not k["test_mode"]
    k = dataset_kwargs(get_config("vidar_1_8_nusc_3future"))
    assert k["rand_frame_interval"] == (-1, 1, 2) and k["voxel_size"] == (1.0,) * 3 and k["future_length"] == 4
    assert dataset_kwargs(get_config("vidar_1_8_nusc_3future"), test_mode=True)["future_length"] == 6
    # the 1/8 stride is a TRAINING-split setting (vidar_1_8_nusc_1future.py:338-342 vs :345-368): evaluation covers the
    # whole val split unless a stride is asked for explicitly
    assert dataset_kwargs(get_config("vidar_1_8_nusc_1future"), test_mode=True)["load_frame_interval"] is None
    assert dataset_kwargs(get_config("vidar_1_8_nusc_1future"), test_mode=True, test_stride=8)["load_frame_interval"] == 8
    assert dataset_kwargs(get_config("vidar_full_nusc_1future"))["load_frame_interval"] == 1
    k = dataset_kwargs(get_config("vidar_OpenScene_mini_full_3future"))
    assert k["dataset"] == "nuplan" and k["rand_frame_interval"] == (1,) and abs(k["img_scale"] - 2 / 3) < 1e-12
    ref = Path("/root/reference/projects/configs/vidar_pretrain/nusc_1_8_subset/vidar_1_8_nusc_3future.py")
    if ref.exists():
        from vidar_amd.plugin.config import Config
        cfg = Config.fromfile(str(ref))
        for name, test in (("vidar_1_8_nusc_3future", False), ("vidar_1_8_nusc_3future", True)):
            assert dataset_kwargs(get_config(name), test, cfg) == dataset_kwargs(get_config(name), test)


def test_hsv_pair_agrees_with_the_standard_library_formula():
    """[3P] cv2 is not installed; the BGR<->HSV pair is the published hexcone model, which Python's `colorsys` implements
    independently: hue / 360, saturation and value agree on random float pixels (both directions), so what stays
    unpinned is only cv2's choice of scale for float images (H in degrees, S in [0, 1], V in the input's scale)"""
    import colorsys
    from vidar_amd.data.augment import bgr2hsv, hsv2bgr
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 255, (40, 3)).astype(np.float32)
    hsv = bgr2hsv(img[None])[0]
    for (b, g, r), (h, s, v) in zip(img.astype(np.float64), hsv.astype(np.float64)):
        hh, ss, vv = colorsys.rgb_to_hsv(r, g, b)
        assert abs(h / 360.0 - hh) < 1e-5 and abs(s - ss) < 1e-5 and abs(v - vv) < 1e-3
        rr, gg, bb = colorsys.hsv_to_rgb(hh, ss, vv)
        back = hsv2bgr(np.array([[[h, s, v]]], np.float32))[0, 0]
        np.testing.assert_allclose(back, [bb, gg, rr], rtol=0, atol=2e-3)
