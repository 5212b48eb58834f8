"""GPU parity tests of the pose stage, stage by stage through the C ABI, against the oracle
(pinned by the reference's own outputs, tests/test_oracle_hpe_golden.py) and the committed goldens.

Tolerances: warp and crop homography are index / float32 work -> bit-exact; decode and the
float64 reconstruction -> 1e-5 (the reference's softmax sums run in float32, numpy pairwise order;
ours accumulate in float64 -- observed difference 2e-6); backbone (bf16 storage, f32 accumulate)
-> within the north star's 1e-3 on 3D joints against the bf16-faithful oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from isbfsar_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def assets():
    a = os.path.join(ROOT, "isbfsar_amd", "assets")
    return np.load(os.path.join(a, "32_to_122.npy")), json.load(open(os.path.join(a, "skeleton_types.json")))


@pytest.fixture(scope="module")
def eng(assets):
    from isbfsar_amd.hpe_engine import HpeEngine
    e = HpeEngine(device=0, max_batch=8)
    e.set_joint_map(assets[0], assets[1]["smpl+head_30"]["indices"])
    yield e
    e.close()


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "hpe_post.npz"))


def _K():
    from oracle import hpe_oracle as ho
    return ho.intrinsics_matrix(384.025146484375, 384.025146484375, 319.09661865234375, 237.75723266601562)


def _abs_amplification(lg, nk, r, W, idx, eps=1e-4, trials=6, seed=0):
    """Conditioning of the absolute reconstruction AT this input (oracle arithmetic only): the largest change of the
    absolute pose per unit change of the decoded predictions, probed with random perturbations of size eps of pred3d
    (heatmap units) and pred2d / 255. The root depth comes out of a 64 x 3 least-squares fit (misc.py:141-176) whose
    gain is ~ depth / 2D spread of the joints: 5-20 on the synthetic heatmaps here."""
    from oracle import hpe_oracle as ho
    p2, p3 = ho.decode(lg)
    in_fov = ho.is_within_fov(p2)
    base = ho.reconstruct_absolute(p2, p3, nk[None, ...], in_fov)
    rng = np.random.default_rng(seed)
    amp = 0.0
    for _ in range(trials):
        d2 = rng.uniform(-eps, eps, p2.shape) * 255.0
        d3 = rng.uniform(-eps, eps, p3.shape)
        pert = ho.reconstruct_absolute(p2 + d2, p3 + d3, nk[None, ...], in_fov)
        amp = max(amp, float(np.abs(pert - base).max()) / eps)
    return amp


def _check_pose(joints_gpu, lg_gpu, lg_ref, nk, r, W, idx, tag="", lg_f32=None, flat_abs=True):
    """3D joints against the oracle (bf16-faithful mode = the product's storage points): the decoded predictions (pred3d
    in heatmap units, pred2d / 255) and the root-centred pose -- what the AR stage consumes, main.py:103 -- at the north
    star's 1e-3, and the ABSOLUTE pose (what estimate() returns, hpe.py:171) at a flat 1e-3 too (flat_abs; default
    precision: fp16 in the two 8x8 stages). flat_abs=False (the chaotic "signal" weights, the plain-bf16
    precision): 1e-3 wherever the reconstruction is conditioned for it, else (its own amplification at this input) x (the
    prediction error), under a FIXED ceiling of 4e-3 so that a regression cannot hide behind its own error.
    lg_f32: logits of the fp32 oracle for the same crop -> the absolute pose must also sit within 1e-3 of the fp32
    definition (BASELINE north star; error budget in DESIGN.md section 4)."""
    from oracle import hpe_oracle as ho
    ref = ho.postprocess(lg_ref, nk, r, W, idx)
    assert ref is not None
    p2g, p3g = ho.decode(lg_gpu)
    p2o, p3o = ho.decode(lg_ref)
    e_p3, e_p2 = float(np.abs(p3g - p3o).max()), float(np.abs(p2g - p2o).max())
    e_pred = max(e_p3, e_p2 / 255.0)
    e_rc = float(np.abs((joints_gpu - joints_gpu[0]) - (ref - ref[0])).max())
    e_abs = float(np.abs(joints_gpu - ref).max())
    amp = _abs_amplification(lg_ref, nk, r, W, idx)
    msg = f"pose{tag}: |d pred|={e_pred:.2e} (2D {e_p2:.3f} px) |d root-centred|={e_rc:.2e} |d absolute|={e_abs:.2e} (reconstruction gain {amp:.1f})"
    e_f32 = None
    if lg_f32 is not None:
        ref32 = ho.postprocess(lg_f32, nk, r, W, idx)
        assert ref32 is not None
        e_f32 = float(np.abs(joints_gpu - ref32).max())
        msg += f" |d absolute vs fp32 definition|={e_f32:.2e}"
    print(msg)
    assert e_p3 < 1e-3 and e_rc < 1e-3          # north star: 3D joints within 1e-3 (heatmap units / root-centred pose)
    assert e_p2 < (0.1 if flat_abs else 0.5)    # the 2D heat-map coordinate, in pixels of the 256 x 256 crop (observed: a few 1e-2)
    if flat_abs:
        assert e_abs < 1e-3, (e_abs, amp, e_pred)
    else:
        assert e_abs < min(max(1e-3, 1.5 * amp * e_pred), 4e-3), (e_abs, amp, e_pred)
    if e_f32 is not None:
        assert e_f32 < 1e-3, e_f32
    return e_abs


def test_crop_params_match_reference(eng, g):
    from oracle import hpe_oracle as ho
    bbs = np.concatenate([g["bboxes"], synth.bboxes(60, seed=5)])
    H, newK, R = [], [], []
    for i in range(0, len(bbs), 8):
        h, k, r = eng.crop_params(bbs[i:i + 8])
        H.append(h); newK.append(k); R.append(r)
    H, newK, R = np.concatenate(H), np.concatenate(newK), np.concatenate(R)
    for i in range(4):     # reference known answers (misc.homography + hpe.py:96)
        np.testing.assert_array_equal(H[i], g[f"hom{i}_H"][0])
        np.testing.assert_allclose(newK[i], g[f"hom{i}_new_K"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(R[i], g[f"hom{i}_R"][0], rtol=0, atol=1e-12)
    n_h_equal = 0
    for i, bb in enumerate(bbs):
        nk, r, h = ho.crop_params(bb, _K())
        np.testing.assert_allclose(newK[i], nk, rtol=1e-13, atol=1e-10)
        np.testing.assert_allclose(R[i], r[0], rtol=0, atol=1e-13)
        np.testing.assert_allclose(H[i], h[0], rtol=2e-7, atol=1e-9)
        n_h_equal += np.array_equal(H[i], h[0])
    assert n_h_equal >= len(bbs) - 2          # f32 rounding of an f64 product: ties are measure-zero


def test_warp_bit_exact_vs_reference(eng, g):
    for i in range(int(g["n_cases"])):
        frame = np.random.default_rng(int(g[f"c{i}_frame_seed"])).integers(0, 256, (480, 640, 3), dtype=np.uint8)
        crop = eng.warp(frame[None], g[f"c{i}_bbox"][None])
        assert _digest(crop) == str(g[f"c{i}_bbone_in_digest"])       # == what the reference fed its backbone engine
        u8 = np.rint(crop * 255.0).astype(np.uint8)
        assert _digest(u8) == str(g[f"c{i}_warp_digest"])
        np.testing.assert_array_equal(u8[0, :32, :32], g[f"c{i}_warp_patch"])


def test_tta_crop_params_and_warp_match_reference(golden_dir):
    """num_aug = 5 (hpe.py:88-100): five parameter sets and five crops per box, against what the reference's own
    estimate() computed (tests/golden/hpe_tta.npz) and, for a batch of boxes, against the oracle; the full forward is
    refused because the reference defines nothing past the crops."""
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    t = np.load(os.path.join(golden_dir, "hpe_tta.npz"))
    n = int(t["num_aug"])
    e = HpeEngine(device=0, max_batch=32)
    try:
        e.set_augmentations(n)
        for i in range(int(t["n_cases"])):
            bb = t[f"t{i}_bbox"][None]
            H, newK, R = e.crop_params(bb)
            assert H.shape == (n, 3, 3)
            np.testing.assert_allclose(H, t[f"t{i}_H"], rtol=2e-7, atol=1e-9)
            np.testing.assert_allclose(newK, t[f"t{i}_new_K"], rtol=1e-13, atol=1e-9)
            np.testing.assert_allclose(R, t[f"t{i}_homo_inv"], rtol=0, atol=1e-12)
            frame = np.random.default_rng(int(t[f"t{i}_frame_seed"])).integers(0, 256, (480, 640, 3), dtype=np.uint8)
            crops = e.warp(frame[None], bb)
            assert crops.shape == (n, 256, 256, 3)
            u8 = np.rint(crops * 255.0).astype(np.uint8)
            if np.array_equal(H, t[f"t{i}_H"]):            # same f32 matrices -> the same pixels, bit for bit
                assert _digest(crops) == str(t[f"t{i}_bbone_in_digest"])
                for k in range(n):
                    assert _digest(u8[k]) == str(t[f"t{i}_warp_digest"][k])
            np.testing.assert_array_equal(u8[:, 96:160:2, 96:160:2], t[f"t{i}_warp_patch"])
        # a batch of boxes: (box, augmentation) order, each against the oracle
        fr = synth.frames(4, seed=77)
        bb = synth.bboxes(4, seed=77)
        H, newK, R = e.crop_params(bb)
        crops = e.warp(fr, bb)
        n_diff = 0
        for b in range(4):
            nk, hi, h = ho.crop_params_aug(bb[b], _K(), n)
            np.testing.assert_allclose(H[b * n:(b + 1) * n], h, rtol=2e-7, atol=1e-9)
            np.testing.assert_allclose(R[b * n:(b + 1) * n], hi, rtol=0, atol=1e-12)
            for k in range(n):
                n_diff += int((crops[b * n + k] != ho.warp(fr[b], H[b * n + k])).any(axis=-1).sum())
        assert n_diff == 0                                 # the warp of the engine's own H matrices is exact
        with pytest.raises(Exception, match="augmentation"):
            e.forward(fr, bb)
        e.set_augmentations(0)
        assert e.crop_params(bb)[0].shape == (4, 3, 3)
    finally:
        e.close()


def test_warp_batch_vs_oracle(eng):
    from oracle import hpe_oracle as ho
    fr = synth.frames(8, seed=40)
    bb = synth.bboxes(8, seed=40)
    bb[0] = (0, 639, 0, 479)            # whole frame
    bb[1] = (600, 639, 440, 479)        # tiny corner box -> most of the crop falls outside the frame
    crops = eng.warp(fr, bb)
    n_diff = 0
    for i in range(8):
        ref = ho.warp(fr[i], ho.crop_params(bb[i], _K())[2][0])
        n_diff += int((crops[i] != ref).any(axis=-1).sum())
    assert n_diff == 0


def test_post_matches_reference_estimate(eng, g, assets):
    from isbfsar_amd.hpe_engine import HpeEngine
    e122 = HpeEngine(device=0, max_batch=8)
    e122.set_joint_map(assets[0], None)
    for i in range(int(g["n_cases"])):
        e = eng if str(g[f"c{i}_skeleton"]) == "30" else e122
        joints, valid, pred = e.post(g[f"c{i}_head_logits"], g[f"c{i}_bbox"][None], want_pred=True)
        assert bool(valid[0]) == bool(g[f"c{i}_valid"])
        if valid[0]:
            np.testing.assert_allclose(joints[0], g[f"c{i}_pose"], rtol=0, atol=1e-5)
    e122.close()


def test_post_batch_vs_oracle(eng, assets):
    from oracle import hpe_oracle as ho
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    rng = np.random.default_rng(77)
    B = 8
    lg = rng.normal(0, 3, (B, 8, 8, 288)).astype(np.float32)
    for b in range(B):
        for j in range(32):
            hh, ww, dd = rng.integers(0, 8, 3)
            lg[b, hh, ww, j] += 10.0
            lg[b, hh, ww, 32 + dd * 32 + j] += 10.0
    lg[3, 0, 0, :] += 40.0              # everything in the corner: out of FOV -> invalid
    bb = synth.bboxes(B, seed=9)
    joints, valid, pred = eng.post(lg, bb, want_pred=True)
    for b in range(B):
        p2, p3 = ho.decode(lg[b:b + 1])
        np.testing.assert_allclose(pred[b, :, :2], p2[0], rtol=0, atol=1e-3)
        np.testing.assert_allclose(pred[b, :, 2:], p3[0], rtol=0, atol=1e-5)
        nk, r, _ = ho.crop_params(bb[b], _K())
        ref = ho.postprocess(lg[b:b + 1], nk, r, W, idx)
        assert bool(valid[b]) == (ref is not None)
        if ref is not None:
            np.testing.assert_allclose(joints[b], ref, rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    assert not valid[3] and valid.sum() >= 6


@pytest.fixture(scope="module")
def bbone_state():
    from isbfsar_amd import effnetv2
    return effnetv2.make_state(0)


@pytest.fixture(scope="module")
def eng_w(eng, bbone_state):
    eng.load_weights(bbone_state)
    return eng


def test_backbone_vs_oracle(eng_w, bbone_state, assets):
    """K2: HIP EfficientNetV2-L (default precision: fp16 storage / f32 accumulate -- the reference's engines are fp16,
    7_create_engines.py:10) against our own CPU definition in its storage-faithful mode (parity unpinned w.r.t. MetrABS:
    the reference has neither source nor weights)."""
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    fr = synth.frames(2, seed=0)
    bb = synth.bboxes(2, seed=0)
    crops = np.stack([ho.warp(fr[i], ho.crop_params(bb[i], _K())[2][0]) for i in range(2)])
    feat, logits = eng_w.backbone(crops)
    o16 = EffNetV2LOracle(bbone_state, "f16")
    f16 = o16.backbone(crops)
    l16 = o16.head(f16)
    scale = float(np.abs(f16).max())
    err_feat = float(np.abs(feat - f16).max())
    err_log = float(np.abs(logits - l16).max())
    print(f"backbone: max|feat|={scale:.3f} err_feat={err_feat:.2e} err_logits={err_log:.2e}")
    assert err_feat < 4e-3 * scale                      # fp16 re-rounding noise through 79 blocks: 5.6e-4 on 0.73 measured, ceiling 5x that
    assert err_log < 4e-3 * float(np.abs(l16).max())    # the f32 pose head adds nothing of its own
    p2_g, p3_g = ho.decode(logits)
    p2_o, p3_o = ho.decode(l16)
    assert np.abs(p3_g - p3_o).max() < 1e-3             # north star: 3D joints within 1e-3
    assert np.abs(p2_g - p2_o).max() < 0.255            # same bound in pixel units (x255)
    # drift against the pure-fp32 definition, documented in DESIGN.md
    o32 = EffNetV2LOracle(bbone_state, "f32")
    l32 = o32.head(o32.backbone(crops))
    p2_f, p3_f = ho.decode(l32)
    drift = float(np.abs(p3_g - p3_f).max())
    print(f"backbone: 3D joint drift vs fp32 definition = {drift:.2e}")
    assert drift < 1e-3                                 # north star: within 1e-3 of the fp32 CPU path


def test_backbone_signal_profile_vs_oracle(eng, assets):
    """The same comparison on weights whose activations CARRY the input (effnetv2.make_state profile "signal": the
    final features differ between frames by as much as their own magnitude, and an error anywhere in the network
    reaches them) and whose pose heatmaps are peaked -- the regime of a trained MetrABS. With the default profile the
    features are 99.7 % frame-independent, which hides input-dependent errors."""
    from isbfsar_amd import effnetv2
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    state = effnetv2.make_state(0, "signal", head_gain=0.5)
    eng.load_weights(state)
    try:
        B = 4
        fr = synth.frames(B, seed=0)
        bb = synth.bboxes(B, seed=5)
        crops = np.stack([ho.warp(fr[i], ho.crop_params(bb[i], _K())[2][0]) for i in range(B)])
        feat, logits = eng.backbone(crops)
        o16 = EffNetV2LOracle(state, "f16")
        f16 = o16.backbone(crops)
        l16 = o16.head(f16)
        scale = float(np.abs(f16).max())
        assert float(np.abs(f16[0] - f16[1]).max()) > 0.2 * scale          # the features do depend on the frame
        err_feat = float(np.abs(feat - f16).max())
        rel_l2 = float(np.linalg.norm(feat - f16) / np.linalg.norm(f16))
        p2_g, p3_g = ho.decode(logits)
        p2_o, p3_o = ho.decode(l16)
        e3 = float(np.abs(p3_g - p3_o).max())
        spread = float(p2_o.max() - p2_o.min())
        print(f"signal profile: max|feat|={scale:.2f} err_feat={err_feat:.2e} rel L2={rel_l2:.2e} |d pred3d|={e3:.2e} "
              f"2D spread={spread:.0f} px")
        assert spread > 20                                   # peaked heatmaps: joints spread over the crop
        # a random 79-block network amplifies a rounding flip like any other perturbation: the same storage noise
        # (fp32 vs bf16 definition: 2 %) shows between two bf16 evaluations that sum in different orders
        assert err_feat < 5e-2 * scale and rel_l2 < 3e-2
        assert e3 < 1e-3                                     # north star on 3D joints (heatmap units)
        joints, valid = eng.forward(fr, bb)
        for b in range(B):
            nk, r, H = ho.crop_params(bb[b], _K())
            ref = ho.postprocess(l16[b:b + 1], nk, r, W, st["smpl+head_30"]["indices"])
            assert bool(valid[b]) == (ref is not None)
            if ref is not None:
                _check_pose(joints[b], logits[b:b + 1], l16[b:b + 1], nk, r, W, st["smpl+head_30"]["indices"], tag=f" signal frame {b}", flat_abs=False)
    finally:
        eng.load_weights(effnetv2.make_state(0))            # the module-scoped engine goes back to the default weights


def test_forward_end_to_end_vs_oracle(eng_w, bbone_state, assets):
    """estimate() for a batch: frames + boxes -> joints, against the oracle chain."""
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = 3
    fr = synth.frames(B, seed=20)
    bb = synth.bboxes(B, seed=20)
    joints, valid = eng_w.forward(fr, bb)
    o16 = EffNetV2LOracle(bbone_state, "f16")
    o32 = EffNetV2LOracle(bbone_state, "f32")
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(B)])
    _, lg_gpu = eng_w.backbone(crops)
    for b in range(B):
        nk, r, H = ho.crop_params(bb[b], _K())
        lg = o16.head(o16.backbone(crops[b:b + 1]))
        ref = ho.postprocess(lg, nk, r, W, idx)
        assert bool(valid[b]) == (ref is not None)
        if ref is not None:
            _check_pose(joints[b], lg_gpu[b:b + 1], lg, nk, r, W, idx, tag=f" frame {b}",
                        lg_f32=o32.head(o32.backbone(crops[b:b + 1])))
    # micro-batching (max_batch=8 here) and the device-pointer path give identical results
    import torch
    j2, v2 = eng_w.forward(torch.from_numpy(fr).cuda(), torch.from_numpy(bb).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(j2.cpu().numpy(), joints) and np.array_equal(v2.cpu().numpy(), valid)


def _pcts(d):
    d = np.sort(np.asarray(d, dtype=np.float64))
    return float(np.median(d)), float(d[min(len(d) - 1, int(np.ceil(0.99 * len(d))) - 1)]), float(d[-1])


N_STAT = 64         # frames behind the "within 1e-3 of the fp32 definition" claims (a max over 8 frames is not a guarantee)


def test_absolute_pose_within_1e3_of_fp32_definition(bbone_state, assets):
    """estimate() RETURNS the absolute pose (hpe.py:171; main.py:102 takes its distance), so it has to sit within the north
    star's 1e-3 of the fp32 path too, not only of the storage-faithful oracle. 64 frames (the first 32 are the frames
    bench.py's parity object uses), default weights, against the fp32 oracle: the default layout (fp16 everywhere = the
    reference's TensorRT precision) at a flat 1e-3 on the MAX over all 64, with the p50 / p99 / max printed; round 3's mixed
    layout and plain bf16 are run beside it on the first 16 frames (mixed: flat 1e-3 too; bf16 must be the worst -- its distance
    to fp32 is a property of bf16 storage: oracle/error_budget.py)."""
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = N_STAT
    fr, bb = synth.frames(B, seed=0), synth.bboxes(B, seed=0)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(B)])
    o32 = EffNetV2LOracle(bbone_state, "f32")
    l32 = np.concatenate([o32.head(o32.backbone(crops[i:i + 16])) for i in range(0, B, 16)])
    errs = {}
    for prec, n in (("f16", B), ("bf16_f16tail", 16), ("bf16", 16)):
        e = HpeEngine(device=0, max_batch=B, precision=prec)
        try:
            e.set_joint_map(W, idx)
            e.load_weights(bbone_state)
            joints, valid = e.forward(fr[:n], bb[:n])
        finally:
            e.close()
        d = []
        for b in range(n):
            nk, r, _ = ho.crop_params(bb[b], _K())
            ref = ho.postprocess(l32[b:b + 1], nk, r, W, idx)
            assert ref is not None and valid[b] == 1
            d.append(float(np.abs(joints[b] - ref).max()))
        errs[prec] = np.array(d)
        p50, p99, mx = _pcts(d)
        print(f"absolute pose vs fp32 definition, default weights, precision {prec}, {n} frames: p50 {p50:.2e} p99 {p99:.2e} max {mx:.2e}")
    assert errs["f16"].max() < 1e-3
    assert errs["bf16_f16tail"].max() < 1e-3
    assert np.median(errs["f16"][:16]) < np.median(errs["bf16"]) and np.median(errs["bf16_f16tail"]) < np.median(errs["bf16"])


def test_signal_profile_fp16_vs_fp32_definition(assets):
    """The "signal" weight profile (activations that carry the input, peaked heat-maps: the regime of a trained MetrABS) is the
    hard one: a random 79-block SiLU network amplifies every rounding. fp16 storage in every stage (the default, the reference's
    own TensorRT precision, 7_create_engines.py:10) has to stay inside the north star's 1e-3 of the FP32 ORACLE here too --
    decoded 3D AND the absolute pose estimate() returns -- on the max over 64 frames, with the per-frame p50 / p99 / max printed
    (the absolute pose goes through a least-squares solve that amplifies heat-map noise: a max over 8 frames moved between 5e-4 and
    9e-4 with rounding details alone). The bf16-carrying layouts, run beside it on the first 8 frames, must be
    the worse ones (1.6e-3 / 9.1e-3 and 2.1e-3 / 6.7e-3)."""
    from isbfsar_amd import effnetv2
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    state = effnetv2.make_state(0, "signal", head_gain=0.5)
    B = N_STAT
    fr, bb = synth.frames(B, seed=0), synth.bboxes(B, seed=0)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(B)])
    o32 = EffNetV2LOracle(state, "f32")
    l32 = np.concatenate([o32.head(o32.backbone(crops[i:i + 16])) for i in range(0, B, 16)])
    p2_32, p3_32 = ho.decode(l32)
    res = {}
    for prec, n in (("f16", B), ("bf16_f16tail", 8), ("bf16", 8)):
        e = HpeEngine(device=0, max_batch=B, precision=prec)
        try:
            e.set_joint_map(W, idx)
            e.load_weights(state)
            joints, valid = e.forward(fr[:n], bb[:n])
            _, lg = e.backbone(crops[:n])
        finally:
            e.close()
        p2, p3 = ho.decode(lg)
        e3 = np.abs(p3 - p3_32[:n]).reshape(n, -1).max(axis=1)
        e2 = float(np.abs(p2 - p2_32[:n]).max())
        ea = []
        for b in range(n):
            nk, r, _ = ho.crop_params(bb[b], _K())
            ref = ho.postprocess(l32[b:b + 1], nk, r, W, idx)
            if ref is not None and valid[b]:
                ea.append(float(np.abs(joints[b] - ref).max()))
        assert len(ea) >= n - n // 8, (len(ea), n)          # (a frame whose joints leave the field of view has no pose to compare)
        res[prec] = (float(e3.max()), e2, max(ea), float(e3[:8].max()))
        a50, a99, amx = _pcts(ea)
        d50, d99, dmx = _pcts(e3)
        print(f"signal profile vs fp32 definition, precision {prec}, {len(ea)} of {n} frames: decoded 3D p50 {d50:.2e} p99 {d99:.2e} max {dmx:.2e}; "
              f"2D {e2:.3f} px; absolute pose p50 {a50:.2e} p99 {a99:.2e} max {amx:.2e}")
    assert res["f16"][0] < 1e-3, res["f16"]                   # decoded 3D (heat-map units): the north star's 1e-3
    assert res["f16"][2] < 1e-3, res["f16"]                   # absolute pose: the same flat 1e-3
    assert res["f16"][3] < 0.5 * res["bf16"][0] and res["f16"][3] < 0.5 * res["bf16_f16tail"][0]      # (same 8 frames)


def test_mixed_precision_vs_its_oracle(bbone_state, assets):
    """isb_hpe_cfg.precision = 3 (round 3's layout: bf16, fp16 in the two 8x8 stages and the 640 -> 1280 convolution)
    against the oracle's matching mode ("bf16")."""
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = 3
    fr, bb = synth.frames(B, seed=22), synth.bboxes(B, seed=22)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(B)])
    o16 = EffNetV2LOracle(bbone_state, "bf16")
    e = HpeEngine(device=0, max_batch=8, precision="bf16_f16tail")
    try:
        e.set_joint_map(W, idx)
        e.load_weights(bbone_state)
        joints, valid = e.forward(fr, bb)
        _, lg_gpu = e.backbone(crops)
    finally:
        e.close()
    for b in range(B):
        nk, r, _ = ho.crop_params(bb[b], _K())
        lg = o16.head(o16.backbone(crops[b:b + 1]))
        assert valid[b] == 1
        _check_pose(joints[b], lg_gpu[b:b + 1], lg, nk, r, W, idx, tag=f" mixed layout frame {b}")


def test_plain_bf16_precision_vs_its_oracle(bbone_state, assets):
    """isb_hpe_cfg.precision = 1 (bf16 everywhere, the round-2 layout) against the oracle's matching mode."""
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = 3
    fr, bb = synth.frames(B, seed=21), synth.bboxes(B, seed=21)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(B)])
    o16 = EffNetV2LOracle(bbone_state, "bf16_plain")
    e = HpeEngine(device=0, max_batch=8, precision="bf16")
    try:
        e.set_joint_map(W, idx)
        e.load_weights(bbone_state)
        joints, valid = e.forward(fr, bb)
        _, lg_gpu = e.backbone(crops)
    finally:
        e.close()
    for b in range(B):
        nk, r, _ = ho.crop_params(bb[b], _K())
        lg = o16.head(o16.backbone(crops[b:b + 1]))
        assert valid[b] == 1
        _check_pose(joints[b], lg_gpu[b:b + 1], lg, nk, r, W, idx, tag=f" plain bf16 frame {b}", flat_abs=False)


def test_select_person_matches_reference(eng, g):
    """detector post-processing on the GPU vs the reference's postprocess_yolo_output + hpe.py:63-79"""
    from oracle import hpe_oracle as ho
    boxes, confs = synth.yolo_outputs()
    bbox, found = eng.select_person(boxes, confs, 0.3)
    np.testing.assert_array_equal(bbox, g["yolo_sel"])
    np.testing.assert_array_equal(found, (g["yolo_sel"][:, 0] >= 0).astype(np.uint8))
    b2, c2 = synth.yolo_outputs(n=9, seed=123)           # more frames against the oracle
    bb2, f2 = eng.select_person(b2, c2, 0.3)
    for i in range(9):
        sel = ho.select_person(b2[i:i + 1], c2[i:i + 1], 640, 480)
        assert (sel is None and not f2[i]) or tuple(bb2[i]) == sel


def test_no_person_gives_invalid_pose(eng_w):
    """estimate() returns None when the detector finds nobody (hpe.py:72-73): a (-1,..) box -> valid 0"""
    fr = synth.frames(2, seed=90)
    bb = synth.bboxes(2, seed=90)
    bb[1] = -1
    joints, valid = eng_w.forward(fr, bb)
    assert valid[0] == 1 and valid[1] == 0 and np.all(joints[1] == 0) and np.isfinite(joints).all()


def test_single_frame_call_vs_oracle(eng_w, bbone_state, assets):
    """The live loop's call: ONE frame. Its SE-gated projections run split-K (hpe_api.cpp conv()), a different f32
    summation order from the batched launch: same tolerance against the oracle, within bf16 re-rounding noise of
    the same frame inside a batch, and reproducible bit for bit."""
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    fr = synth.frames(2, seed=31)
    bb = synth.bboxes(2, seed=31)
    j1, v1 = eng_w.forward(fr[:1], bb[:1])
    j1b, _ = eng_w.forward(fr[:1], bb[:1])
    j2, v2 = eng_w.forward(fr, bb)
    assert np.array_equal(j1, j1b) and v1[0] == v2[0] == 1
    nk, r, H = ho.crop_params(bb[0], _K())
    o16 = EffNetV2LOracle(bbone_state, "f16")
    crop = ho.warp(fr[0], H[0])[None]
    _, lg_gpu = eng_w.backbone(crop)                     # B = 1: the same split-K path as the forward call above
    _check_pose(j1[0], lg_gpu, o16.head(o16.backbone(crop)), nk, r, W, idx, tag=" single frame")
    np.testing.assert_allclose(j1[0] - j1[0][0], j2[0] - j2[0][0], rtol=0, atol=1e-3)


def test_shard_invariance(eng_w):
    """Frames are independent units: any split of a batch into shards of two or more frames (what DP sharding
    across GPUs does) gives bit-identical poses (SURVEY.md 8e correctness check). Single-frame calls take the
    split-K projections -- test_single_frame_call_vs_oracle."""
    fr = synth.frames(5, seed=70)
    bb = synth.bboxes(5, seed=70)
    j_all, v_all = eng_w.forward(fr, bb)
    j_a, v_a = eng_w.forward(fr[:2], bb[:2])
    j_b, v_b = eng_w.forward(fr[2:], bb[2:])
    assert np.array_equal(np.concatenate([j_a, j_b]), j_all) and np.array_equal(np.concatenate([v_a, v_b]), v_all)


def test_two_lane_split_is_bit_identical(bbone_state, assets):
    """A batch >= 64 runs as two halves on two streams (hpe_api.cpp lanes); every frame is independent,
    so the poses equal those of the same frames pushed through one lane in small batches."""
    from isbfsar_amd.hpe_engine import HpeEngine
    e = HpeEngine(device=0, max_batch=80)
    try:
        e.set_joint_map(assets[0], assets[1]["smpl+head_30"]["indices"])
        e.load_weights(bbone_state)
        fr = synth.frames(70, seed=300)
        bb = synth.bboxes(70, seed=300)
        j_all, v_all = e.forward(fr, bb)                       # 35 + 35 on two lanes
        parts = [e.forward(fr[i:i + 14], bb[i:i + 14]) for i in range(0, 70, 14)]   # one lane
        assert np.array_equal(np.concatenate([p[0] for p in parts]), j_all)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), v_all)
        assert v_all.sum() > 0
    finally:
        e.close()


def test_single_frame_fusions_are_bit_identical(bbone_state, assets, monkeypatch):
    """The single-frame path folds the squeeze-excite FCs into its neighbours (FC1 into the depthwise launch, FC2 into
    the split-K projection) with the summation order of the stand-alone kernels: with the FC2 fusion switched off
    (ISB_FUSE_SE=0, read when the engine is created) the poses are the same bits."""
    from isbfsar_amd.hpe_engine import HpeEngine
    fr = synth.frames(3, seed=41)
    bb = synth.bboxes(3, seed=41)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("ISB_FUSE_SE", flag)
        e = HpeEngine(device=0, max_batch=4)
        try:
            e.set_joint_map(assets[0], assets[1]["smpl+head_30"]["indices"])
            e.load_weights(bbone_state)
            outs.append([e.forward(fr[i:i + 1], bb[i:i + 1])[0] for i in range(3)])
        finally:
            e.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.isfinite(np.concatenate(outs[0])).all()


def test_full_size_properties(bbone_state, assets):
    """BASELINE configs[1] size (256 frames per GPU): properties that need no oracle run --
    determinism, invariance under re-batching (4 x 64 through two lanes each vs 256 through two lanes),
    a permutation of the frames permutes the poses, all poses finite, every synthetic frame valid."""
    from isbfsar_amd.hpe_engine import HpeEngine
    e = HpeEngine(device=0, max_batch=256)
    try:
        e.set_joint_map(assets[0], None)
        e.load_weights(bbone_state)
        fr = synth.frames(256, seed=4000)
        bb = synth.bboxes(256, seed=4000)
        j1, v1 = e.forward(fr, bb)
        j2, v2 = e.forward(fr, bb)
        assert np.array_equal(j1, j2) and np.array_equal(v1, v2)
        parts = [e.forward(fr[i:i + 64], bb[i:i + 64]) for i in range(0, 256, 64)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), j1)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), v1)
        perm = np.random.default_rng(0).permutation(256)
        jp, vp = e.forward(fr[perm], bb[perm])
        assert np.array_equal(jp, j1[perm]) and np.array_equal(vp, v1[perm])
        assert np.isfinite(j1).all() and j1.shape == (256, 122, 3)
        assert v1.all()
    finally:
        e.close()


def test_hpe_error_behaviour(bbone_state, assets):
    """The boundary's error contract: every failure is a negative return code + isb_last_error text,
    surfaced by the Python layer as IsbError; nothing is computed on a half-configured handle."""
    from isbfsar_amd import _lib
    from isbfsar_amd.hpe_engine import HpeEngine
    e = HpeEngine(device=0, max_batch=4)
    try:
        fr = synth.frames(2, seed=1)
        bb = synth.bboxes(2, seed=1)
        with pytest.raises(_lib.IsbError, match="weights"):
            e.forward(fr, bb)
        bad = dict(bbone_state)
        bad.pop(next(k for k in bad if k.endswith(".w")))
        with pytest.raises(_lib.IsbError, match="missing"):
            e.load_weights(bad)
        with pytest.raises(_lib.IsbError):
            e.load_weights(b"not a blob at all")
        e.load_weights(bbone_state)
        with pytest.raises(_lib.IsbError, match="joint map"):
            e.forward(fr, bb)
        with pytest.raises(_lib.IsbError, match="outside"):
            e.set_joint_map(assets[0], [0, 1, 200])
        e.set_joint_map(assets[0], None)
        with pytest.raises(_lib.IsbError, match="max_batch"):
            e.crop_params(synth.bboxes(5, seed=2))             # stage hooks run one micro-batch
        j, v = e.forward(synth.frames(5, seed=2), synth.bboxes(5, seed=2))      # the full stage micro-batches (4 + 1)
        assert j.shape == (5, 122, 3) and np.isfinite(j).all()
        with pytest.raises(ValueError):
            e.forward(fr[:, :100], bb)                         # wrong frame size is caught before the library
    finally:
        e.close()


def test_pose_windows_kernel():
    import torch
    from isbfsar_amd.hpe_engine import pose_windows
    rng = np.random.default_rng(3)
    j = rng.normal(0, 1, (3, 20, 30, 3)).astype(np.float32)
    w = pose_windows(torch.from_numpy(j).cuda(), 16).cpu().numpy()
    assert w.shape == (3 * 5, 16, 90)
    c = j - j[:, :, :1, :]
    for cam in range(3):
        for k in range(5):
            assert np.array_equal(w[cam * 5 + k], c[cam, k:k + 16].reshape(16, 90))


def test_pose_distance_kernel():
    """main.py:102: np.sqrt(np.sum(np.square(np.array([0, 0, 0]) - np.array(pose[0])))) * 2.5 on the float64 pose."""
    import torch
    from isbfsar_amd.hpe_engine import pose_distance
    rng = np.random.default_rng(5)
    j = (rng.normal(0, 1, (4, 7, 30, 3)) * np.array([0.4, 0.4, 1.0]) + np.array([0.0, 0.0, 2.5])).astype(np.float32)
    d = pose_distance(torch.from_numpy(j).cuda()).cpu().numpy()
    assert d.shape == (4, 7)
    pose = j.astype(np.float64)
    ref = np.sqrt(np.sum(np.square(np.array([0, 0, 0]) - pose[..., 0, :]), axis=-1)) * 2.5
    assert np.array_equal(d, ref.astype(np.float32))


def test_human_pose_estimator_dropin(bbone_state, assets):
    from isbfsar_amd.modules.hpe.hpe import HumanPoseEstimator
    from isbfsar_amd.params import MetrabsHIPConfig, RealSenseIntrinsics
    cfg = MetrabsHIPConfig()
    cfg.weights = bbone_state
    cfg.fixed_bbox = (192, 448, 48, 432)
    cfg.max_batch = 2
    est = HumanPoseEstimator(cfg, RealSenseIntrinsics())
    frame = synth.frames(1, seed=0)[0]
    res = est.estimate(frame)
    assert res is not None
    assert res["pose"].shape == (30, 3) and res["pose"].dtype == np.float64
    assert len(res["edges"]) == 29 and res["bbox"] == (192, 448, 48, 432)
    # main.py:102-105 consumes it like this
    pose = res["pose"] - res["pose"][0, :]
    assert pose.reshape(-1).shape == (90,)
    jb = HumanPoseEstimator(cfg, RealSenseIntrinsics(), just_box=True)
    assert jb.estimate(frame) == {"bbox": (192, 48, 448, 432)}
    nob = HumanPoseEstimator(cfg, RealSenseIntrinsics(), just_box=True, bbox_provider=lambda f: None)
    assert nob.estimate(frame) is None
    # with a detector callable (anything that returns the YOLOv4 export tensors, hpe.py:59-60) the
    # reference flow detector -> post-processing -> crop runs end to end on the GPU
    boxes, confs = synth.yolo_outputs()
    det = HumanPoseEstimator(cfg, RealSenseIntrinsics(), detector=lambda f: (boxes[0:1], confs[0:1]))
    r2 = det.estimate(frame)
    assert r2 is not None and r2["bbox"] == (269, 459, 280, 455)
    nodet = HumanPoseEstimator(cfg, RealSenseIntrinsics(), detector=lambda f: (boxes[3:4], confs[3:4]))
    assert nodet.estimate(frame) is None
    # num_aug > 0 (params.py:36 disables it): the constructor succeeds like the reference's, the augmented crops are
    # available stage by stage, and estimate() raises like the reference's (hpe.py:108) -- with an explanation
    cfg.num_aug = 5
    cfg.max_batch = 8
    tta = HumanPoseEstimator(cfg, RealSenseIntrinsics())
    assert tta.n_test == 5
    assert tta.engine.warp(frame[None], np.array([cfg.fixed_bbox], np.int32)).shape == (5, 256, 256, 3)
    with pytest.raises(Exception, match="augmentation"):
        tta.estimate(frame)


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_large_micro_batches_are_bit_identical(bbone_state, assets, monkeypatch, lanes):
    """bench.py's 2048-frame line runs micro-batches of 1024 frames (kMaxMicroBatch), i.e. tensors of up to
    2^31 bytes where the 32-bit-offset kernels hand over to others. Frames are independent, so 1024-frame micro-batches
    (one lane: 1024-frame tensors; two lanes: 512) must give the bits of 256-frame micro-batches."""
    import torch
    from isbfsar_amd.hpe_engine import HpeEngine
    monkeypatch.setenv("ISB_HPE_LANES", lanes)
    n = 1024
    fr = torch.from_numpy(synth.frames(64, seed=400)).cuda().repeat(n // 64, 1, 1, 1)
    # 16 copies of 64 frames, each copy with its own boxes: every frame of the batch is a different crop
    bb = torch.from_numpy(np.concatenate([synth.bboxes(64, seed=401 + i) for i in range(n // 64)])).cuda()
    outs = []
    for mb in (1024, 256):
        e = HpeEngine(device=0, max_batch=mb)
        try:
            e.set_joint_map(assets[0], None)
            e.load_weights(bbone_state)
            j, v = e.forward(fr, bb)
            torch.cuda.synchronize()
            outs.append((j.cpu().numpy(), v.cpu().numpy()))
        finally:
            e.close()
    assert outs[0][1].sum() > n // 2
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_roi_only_host_input_is_bit_identical(bbone_state, assets, monkeypatch):
    """With frames in device-mapped pinned host memory isb_hpe_forward_host pulls only the source rectangle each crop can
    reach (one gather kernel over PCIe, ISB_HPE_ROI=1) instead of copying whole 921 600-byte frames: same bits as whole
    frames (ISB_HPE_ROI=0), as pageable frames and as the device-pointer entry, also for boxes that touch or leave the frame
    (the crop's pre-image is then clipped to the frame)."""
    import torch
    from isbfsar_amd.hpe_engine import HpeEngine
    n = 12
    fr = synth.frames(n, seed=500)
    bb = synth.bboxes(n, seed=500)
    bb[0] = (0, 200, 0, 300)            # x1, x2, y1, y2: top-left corner of the frame
    bb[1] = (440, 639, 180, 479)        # bottom-right corner
    bb[2] = (0, 639, 0, 479)            # the whole frame
    bb[3] = (600, 639, 10, 60)          # a sliver at the right edge
    fr_pinned = torch.from_numpy(fr).pin_memory().numpy()
    outs = {}
    for m in ("1", "0"):
        monkeypatch.setenv("ISB_HPE_ROI", m)
        e = HpeEngine(device=0, max_batch=8)       # micro-batches of 8 + 4: the descriptors are indexed per micro-batch
        try:
            e.set_joint_map(assets[0], assets[1]["smpl+head_30"]["indices"])
            e.load_weights(bbone_state)
            outs[m] = e.forward(fr_pinned, bb)
            if m == "1":
                outs["pageable"] = e.forward(fr, bb)
                j_dev, v_dev = e.forward(torch.from_numpy(fr).cuda(), torch.from_numpy(bb).cuda())
                torch.cuda.synchronize()
                outs["dev"] = (j_dev.cpu().numpy(), v_dev.cpu().numpy())
        finally:
            e.close()
    for k in ("0", "pageable", "dev"):
        assert np.array_equal(outs["1"][0], outs[k][0]) and np.array_equal(outs["1"][1], outs[k][1]), k
    assert outs["1"][1].sum() >= n - 4


def test_submitted_host_batches_are_bit_identical(bbone_state, assets):
    """isb_hpe_submit_host / isb_hpe_wait_host: two host batches in flight (batch k + 1's frames cross PCIe on the copy engine
    while batch k computes, the lanes run on without a drain) give the bits of the synchronous isb_hpe_forward_host, in
    submission order; a third submit finishes the oldest first; pageable frames work (the submit then blocks for the copy);
    the synchronous entry may be mixed in; wait without a submission is an ISB_ERR_STATE. Micro-batches: max_batch 8 < 12."""
    import torch
    from isbfsar_amd import _lib
    from isbfsar_amd.hpe_engine import HpeEngine
    n = 12
    frs = [synth.frames(n, seed=600 + i) for i in range(4)]
    bbs = [synth.bboxes(n, seed=600 + i) for i in range(4)]
    pinned = [torch.from_numpy(f).pin_memory().numpy() for f in frs]
    e = HpeEngine(device=0, max_batch=8)
    try:
        e.set_joint_map(assets[0], assets[1]["smpl+head_30"]["indices"])
        e.load_weights(bbone_state)
        want = [e.forward(pinned[i], bbs[i]) for i in range(4)]
        got = []
        e.submit(pinned[0], bbs[0])
        for i in range(1, 4):                                # submit batch i, then collect batch i - 1
            e.submit(pinned[i], bbs[i])
            got.append(e.wait())
        mixed = e.forward(pinned[1], bbs[1])                 # the synchronous entry with batch 3 still outstanding
        got.append(e.wait())
        for (j, v), (jw, vw) in zip(got, want):
            assert np.array_equal(j, jw) and np.array_equal(v, vw)
        assert np.array_equal(mixed[0], want[1][0]) and np.array_equal(mixed[1], want[1][1])
        with pytest.raises(RuntimeError):
            e.wait()
        with pytest.raises(_lib.IsbError, match="no submission"):
            _lib.check(_lib.lib().isb_hpe_wait_host(e._h), "isb_hpe_wait_host")
        # straight through the C entry: a third submit completes the oldest (its arrays are written at that moment)
        outs = [(np.zeros((n, e.n_out, 3), np.float32), np.zeros((n,), np.uint8)) for _ in range(3)]
        for i in range(3):
            _lib.check(_lib.lib().isb_hpe_submit_host(e._h, pinned[i].ctypes.data, bbs[i].ctypes.data, n, outs[i][0].ctypes.data,
                                                      outs[i][1].ctypes.data), "isb_hpe_submit_host")
        assert np.array_equal(outs[0][0], want[0][0]) and np.array_equal(outs[0][1], want[0][1])
        for i in (1, 2):
            _lib.check(_lib.lib().isb_hpe_wait_host(e._h), "isb_hpe_wait_host")
            assert np.array_equal(outs[i][0], want[i][0]) and np.array_equal(outs[i][1], want[i][1])
        e.submit(pinned[0], bbs[0])                          # the handle's configuration is frozen while a batch is outstanding
        with pytest.raises(_lib.IsbError, match="outstanding"):
            e.set_joint_map(assets[0], None)
        e.wait()
        e.submit(frs[3], bbs[3])                             # pageable frames
        j, v = e.wait()
        assert np.array_equal(j, want[3][0]) and np.array_equal(v, want[3][1])
        assert want[0][1].sum() > 0
    finally:
        e.close()


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_fused_8x8_chain_is_bit_identical_to_the_five_launch_path(bbone_state, assets, monkeypatch, precision):
    """conv_mb8.hip: the 31 stride-1 MBConv blocks of the two 8 x 8 stages as ONE launch (a workgroup owns a sample for the whole
    chain: expand from register-streamed weights, depthwise from LDS, squeeze-excite in the workgroup, gated projection, residual
    in LDS). Every sum keeps the order of the five-launch path, so features and poses are the same BITS. The chain measured 2x
    SLOWER than the five launches (EXPERIMENTS.md round 4: per-sample weight streams are bound by L2 -> CU delivery), so it is off by
    default and ISB_MB8=1 (read when the engine is created) selects it. Round 5: the kernel is compiled in PROBE builds only
    (ISB_BUILD_PROBES=1), and its depthwise taps are v_dot2 sums, so the path it is compared with runs the
    v_dot2 depthwise kernel too (ISB_DWMM=0, ISB_MBF8=0: the product's batch path now puts the taps on the matrix pipe)."""
    import ctypes
    from isbfsar_amd import _lib
    from isbfsar_amd.hpe_engine import HpeEngine
    if not ctypes.CDLL(_lib.LIB_PATH).isbfsar_probe_build():
        pytest.skip("mb8_chain_kernel is compiled in probe builds only (ISB_BUILD_PROBES=1 python -m isbfsar_amd.build --force)")
    monkeypatch.setenv("ISB_DWMM", "0")
    monkeypatch.setenv("ISB_MBF8", "0")
    from oracle import hpe_oracle as ho
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = 70                                  # above the chain's batch threshold, not a multiple of anything
    fr, bb = synth.frames(B, seed=3), synth.bboxes(B, seed=3)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(4)])
    crops = np.concatenate([crops] * 18)[:B]
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ISB_MB8", flag)
        e = HpeEngine(device=0, max_batch=128, precision=precision)
        try:
            e.set_joint_map(W, idx)
            e.load_weights(bbone_state)
            feat, logits = e.backbone(crops)
            joints, valid = e.forward(fr, bb)
            again, _ = e.forward(fr, bb)
        finally:
            e.close()
        assert np.array_equal(joints, again)
        res[flag] = (feat, logits, joints, valid)
    assert np.isfinite(res["1"][0]).all() and float(np.abs(res["1"][0]).max()) > 0
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_fused_front_of_the_8x8_blocks_is_bit_identical(bbone_state, assets, monkeypatch, precision):
    """conv_mb8.hip mbfront8_kernel: expand 1x1 + SiLU + depthwise 3x3 + SiLU + squeeze-excite pool of the stride-1 MBConv blocks with
    384 input channels on 8 x 8 maps in ONE launch (weights stationary in registers, the expanded tensor never leaves the chip, the
    depthwise stage wave-local): features and poses are the bits of the expand-GEMM + depthwise-kernel path (ISB_MBF8=0/1, read when
    the engine is created)."""
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = 45                                  # above the batch threshold; more samples than sample sequences share out evenly
    fr, bb = synth.frames(B, seed=4), synth.bboxes(B, seed=4)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(3)])
    crops = np.concatenate([crops] * 15)[:B]
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ISB_MBF8", flag)
        e = HpeEngine(device=0, max_batch=64, precision=precision)
        try:
            e.set_joint_map(W, idx)
            e.load_weights(bbone_state)
            feat, logits = e.backbone(crops)
            joints, valid = e.forward(fr, bb)
        finally:
            e.close()
        res[flag] = (feat, logits, joints, valid)
    assert np.isfinite(res["1"][0]).all() and float(np.abs(res["1"][0]).max()) > 0
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_fused_front_of_the_16x16_blocks_is_bit_identical(bbone_state, assets, monkeypatch, precision):
    """conv_mb16.hip mbfront16_kernel (round 5): expand 1x1 + SiLU + depthwise 3x3 + SiLU + squeeze-excite pool of the 28 stride-1
    MBConv blocks on 16 x 16 maps in ONE launch, a sample walked in bands of two image rows (ring of six expanded rows per wave,
    depthwise taps on the matrix pipe): features and poses are the bits of the expand-GEMM + dwconv3x3_mm_kernel path (ISB_MBF16=0/1,
    read when the engine is created). B = 45: above the batch threshold, more samples than sample sequences share out evenly."""
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    W, st = assets
    idx = st["smpl+head_30"]["indices"]
    B = 45
    fr, bb = synth.frames(B, seed=5), synth.bboxes(B, seed=5)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], _K())[2][0]) for b in range(3)])
    crops = np.concatenate([crops] * 15)[:B]
    res = {}
    monkeypatch.setenv("ISB_DWMM16", "1")              # the reference path keeps the fused front's depthwise arithmetic
    for flag in ("1", "0"):
        monkeypatch.setenv("ISB_MBF16", flag)
        e = HpeEngine(device=0, max_batch=64, precision=precision)
        try:
            e.set_joint_map(W, idx)
            e.load_weights(bbone_state)
            feat, logits = e.backbone(crops)
            joints, valid = e.forward(fr, bb)
            again, _ = e.forward(fr, bb)
        finally:
            e.close()
        assert np.array_equal(joints, again)
        res[flag] = (feat, logits, joints, valid)
    assert np.isfinite(res["1"][0]).all() and float(np.abs(res["1"][0]).max()) > 0
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)


def test_set_lanes_changes_no_bit_and_refuses_nonsense(bbone_state, assets):
    """isb_hpe_set_lanes (round 5): 1 = whole-batch launches on the caller's stream (for callers that keep several batches in flight on
    several engines), 2 = the default split into two half-batch lanes. Frames are independent: the same bits either way; lane counts
    outside 1..4 are refused with ISB_ERR_INVALID."""
    from isbfsar_amd._lib import IsbError
    from isbfsar_amd.hpe_engine import HpeEngine
    e = HpeEngine(device=0, max_batch=72)
    try:
        e.set_joint_map(assets[0], assets[1]["smpl+head_30"]["indices"])
        e.load_weights(bbone_state)
        fr = synth.frames(70, seed=301)
        bb = synth.bboxes(70, seed=301)
        j2, v2 = e.forward(fr, bb)                              # 35 + 35 on two lanes
        e.set_lanes(1)
        j1, v1 = e.forward(fr, bb)                              # one 70-frame lane
        assert np.array_equal(j1, j2) and np.array_equal(v1, v2) and v1.sum() > 0
        for bad in (0, 5, -1):
            with pytest.raises(IsbError):
                e.set_lanes(bad)
        e.set_lanes(2)
        j2b, _ = e.forward(fr, bb)
        assert np.array_equal(j2b, j2)
    finally:
        e.close()


def test_shared_engines_read_one_model_and_give_the_same_bits(bbone_state, assets):
    """isb_hpe_create_shared: engines that keep several batches in flight share ONE device copy of the weights. Three
    engines of a family (parent + two children), each with its own batch on its own stream at the same time, give bit for bit what the
    parent gives batch after batch; the family's device memory is one model + one workspace per engine; a child refuses weights and
    joint maps of its own; the model outlives the parent."""
    import torch
    from isbfsar_amd._lib import IsbError
    from isbfsar_amd.hpe_engine import HpeEngine
    W, st = assets
    B = 40
    parent = HpeEngine(device=0, max_batch=B)
    parent.set_joint_map(W, st["smpl+head_30"]["indices"])
    parent.load_weights(bbone_state)
    parent.set_lanes(1)
    kids = [parent.share(), parent.share()]
    try:
        batches = [(torch.from_numpy(synth.frames(B, seed=900 + k)).cuda(), torch.from_numpy(synth.bboxes(B, seed=900 + k)).cuda()) for k in range(3)]
        ref = []
        for f, b in batches:                                    # one at a time through the parent
            j, v = parent.forward(f, b)
            torch.cuda.synchronize()
            ref.append((j.cpu().numpy(), v.cpu().numpy()))
        assert all(v.sum() > 0 for _, v in ref)
        streams = [torch.cuda.Stream() for _ in range(3)]
        outs = []
        torch.cuda.synchronize()
        for e, s, (f, b) in zip([parent] + kids, streams, batches):     # all three in flight
            with torch.cuda.stream(s):
                outs.append(e.forward(f, b))
        torch.cuda.synchronize()
        for (j, v), (jr, vr) in zip(outs, ref):
            assert np.array_equal(j.cpu().numpy(), jr) and np.array_equal(v.cpu().numpy(), vr)
        m0, w0, n0 = parent.memory()
        assert n0 == 3 and m0 > 200e6                           # EfficientNetV2-L: ~240 MB of folded 16-bit weights + packed copies
        for k in kids:
            m, w, n = k.memory()
            assert m == m0 and n == 3 and 0 < w <= w0 * 1.01    # the same model, a workspace of its own (same batch size)
        free_before = torch.cuda.mem_get_info()[0]
        extra = parent.share()
        j, _ = extra.forward(*batches[0])
        torch.cuda.synchronize()
        used = free_before - torch.cuda.mem_get_info()[0]
        assert used < w0 + 64e6 and used < m0, (used, w0, m0)   # a fourth engine costs a workspace, not a model
        assert np.array_equal(j.cpu().numpy(), ref[0][0])
        extra.close()
        with pytest.raises(IsbError):
            kids[0].load_weights(bbone_state)
        with pytest.raises(IsbError):
            kids[0].set_joint_map(W, None)
        parent.close()                                          # the children keep the model alive
        j, v = kids[1].forward(*batches[2])
        torch.cuda.synchronize()
        assert np.array_equal(j.cpu().numpy(), ref[2][0]) and kids[1].memory()[2] == 2
    finally:
        for k in kids:
            k.close()
        parent.close()
