"""CPU: oracle/hpe_oracle.py against outputs of the reference's own code (misc.homography, the torch
ImageTransformer, HumanPoseEstimator.estimate through a fake Runner) -- SURVEY.md 8c G4-G6."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import hpe_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "hpe_post.npz"))


@pytest.fixture(scope="module")
def assets():
    a = os.path.join(ROOT, "isbfsar_amd", "assets")
    W = np.load(os.path.join(a, "32_to_122.npy"))
    st = json.load(open(os.path.join(a, "skeleton_types.json")))
    return W, st


K = ho.intrinsics_matrix(384.025146484375, 384.025146484375, 319.09661865234375, 237.75723266601562)


def test_homography_known_answers(g):
    for i, bbox in enumerate(g["bboxes"]):
        new_K, R, H = ho.crop_params(bbox, K)
        np.testing.assert_allclose(new_K, g[f"hom{i}_new_K"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(R, g[f"hom{i}_R"], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(H, g[f"hom{i}_H"])
    # SURVEY 8c probe: homography(200,440,60,420) -> focal ~273.0926, principal point (128,128)
    new_K, _, _ = ho.crop_params((200, 440, 60, 420), K)
    assert abs(new_K[0, 0] - 273.0926) < 1e-3 and new_K[0, 2] == 128 and new_K[1, 2] == 128


def test_warp_bit_exact(g):
    for i in range(int(g["n_cases"])):
        frame = np.random.default_rng(int(g[f"c{i}_frame_seed"])).integers(0, 256, (480, 640, 3), dtype=np.uint8)
        _, _, H = ho.crop_params(g[f"c{i}_bbox"], K)
        np.testing.assert_array_equal(H[0], g[f"c{i}_H"][0])
        crop = ho.warp(frame, H[0])
        assert crop.dtype == np.float32 and crop.shape == (256, 256, 3)
        assert _digest(crop[None]) == str(g[f"c{i}_bbone_in_digest"])          # == what the backbone engine was fed
        u8 = np.rint(crop * 255.0).astype(np.uint8)
        assert _digest(u8[None]) == str(g[f"c{i}_warp_digest"])
        np.testing.assert_array_equal(u8[:32, :32], g[f"c{i}_warp_patch"])
        np.testing.assert_array_equal(u8[112:144, 112:144], g[f"c{i}_warp_center"])


def test_tta_tables_and_crops(golden_dir):
    """Test-time augmentation (hpe.py:88-100, misc.py:312-327) as far as the reference executes it: the tables of
    get_augmentations, per-augmentation new_K / homo_inv / H and the five warped crops the reference's own
    estimate() produced with num_aug = 5 (tests/golden/hpe_tta.npz, oracle/gen_golden_hpe.py::gen_tta). The oracle
    restatement AND the product's host-side mirror of the tables are pinned to them."""
    from isbfsar_amd.hpe_engine import get_augmentations as product_aug
    t = np.load(os.path.join(golden_dir, "hpe_tta.npz"))
    n = int(t["num_aug"])
    for fn in (ho.get_augmentations, product_aug):
        flip, rotflip, gammas, scales = fn(n)
        np.testing.assert_array_equal(flip, t["aug_should_flip"])
        np.testing.assert_array_equal(rotflip, t["aug_rotflip"])
        np.testing.assert_array_equal(gammas, t["aug_gammas"])
        np.testing.assert_array_equal(scales, t["aug_scales"])
        assert str(rotflip.dtype) == str(t["aug_rotflip_dtype"])
    for i in range(int(t["n_cases"])):
        new_K, homo_inv, H = ho.crop_params_aug(t[f"t{i}_bbox"], K, n)
        np.testing.assert_array_equal(H, t[f"t{i}_H"])
        np.testing.assert_allclose(new_K, t[f"t{i}_new_K"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(homo_inv, t[f"t{i}_homo_inv"], rtol=0, atol=1e-12)
        frame = np.random.default_rng(int(t[f"t{i}_frame_seed"])).integers(0, 256, (480, 640, 3), dtype=np.uint8)
        crops = np.stack([ho.warp(frame, H[k]) for k in range(n)])
        assert _digest(crops) == str(t[f"t{i}_bbone_in_digest"])               # what the backbone engine was handed
        u8 = np.rint(crops * 255.0).astype(np.uint8)
        for k in range(n):
            assert _digest(u8[k]) == str(t[f"t{i}_warp_digest"][k])
        np.testing.assert_array_equal(u8[:, 96:160:2, 96:160:2], t[f"t{i}_warp_patch"])
        assert "Error" in str(t[f"t{i}_after_warp"])       # the reference itself cannot go past the crops (hpe.py:108)


def test_postprocess_matches_estimate(g, assets):
    W, st = assets
    n_valid = 0
    for i in range(int(g["n_cases"])):
        new_K, R, _ = ho.crop_params(g[f"c{i}_bbox"], K)
        idx = st["smpl+head_30"]["indices"] if str(g[f"c{i}_skeleton"]) == "30" else None
        pose = ho.postprocess(g[f"c{i}_head_logits"], new_K, R, W, idx)
        if not bool(g[f"c{i}_valid"]):
            assert pose is None                                                # hpe.py:152-153
            continue
        n_valid += 1
        np.testing.assert_allclose(pose, g[f"c{i}_pose"], rtol=0, atol=1e-9)
        assert pose.shape == ((30, 3) if idx is not None else (122, 3))
    assert n_valid >= 4


def test_skeleton_assets(assets):
    W, st = assets
    assert W.shape == (32, 122) and np.allclose(W.sum(0), 1.0, atol=1e-5)
    s = st["smpl+head_30"]
    assert s["indices"] == [23] + list(range(23)) + [28, 35, 41, 72, 83, 89] and len(s["edges"]) == 29


def test_root_centre():
    pose = np.arange(90, dtype=np.float64).reshape(30, 3) / 10 + 1
    d, flat = ho.root_centre(pose)
    assert flat.shape == (90,) and np.all(flat[:3] == 0)
    assert abs(d - np.linalg.norm(pose[0]) * 2.5) < 1e-12


def test_detector_postprocess_matches_reference(g):
    """G8: top-person selection (hpe.py:59-79 + misc.postprocess_yolo_output/nms_cpu) on seeded YOLO tensors."""
    from isbfsar_amd import synth
    boxes, confs = synth.yolo_outputs()
    assert _digest(boxes) == str(g["yolo_boxes_digest"]) and _digest(confs) == str(g["yolo_confs_digest"])
    for i in range(boxes.shape[0]):
        sel = ho.select_person(boxes[i:i + 1], confs[i:i + 1], 640, 480)
        want = tuple(int(v) for v in g["yolo_sel"][i])
        assert (sel is None and want[0] < 0) or sel == want
    assert (g["yolo_sel"][:, 0] < 0).sum() == 2
