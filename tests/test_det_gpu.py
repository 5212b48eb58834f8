"""GPU tests of the YOLOv4 person detector (isb_det_*) stage by stage against its CPU definition
(oracle/yolov4_oracle.py -- PARITY UNPINNED: the reference has neither the network's definition nor its weights; what the
reference pins is the input / output contract, hpe.py:51-60, and the post-processing, tested in test_hpe_gpu.py)."""
import numpy as np
import pytest

from isbfsar_amd import synth, yolov4

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def state():
    return yolov4.make_state(0)


@pytest.fixture(scope="module")
def det(state):
    from isbfsar_amd.det_engine import DetEngine
    d = DetEngine(device=0, max_batch=4)
    d.load_weights(state)
    yield d
    d.close()


def _structured_frames(n, seed):
    """frames with large-scale structure (blobs) besides noise: the area resize and the network see more than white noise"""
    out = synth.frames(n, seed=seed)
    yy, xx = np.mgrid[0:480, 0:640]
    for i in range(n):
        rng = np.random.default_rng(seed + 77 + i)
        img = out[i].astype(np.float32) * 0.3
        for _ in range(5):
            cx, cy, s = rng.uniform(0, 640), rng.uniform(0, 480), rng.uniform(30, 120)
            img += np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))[..., None] * rng.uniform(0, 200, 3)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def test_preprocess_matches_oracle(det):
    from oracle.yolov4_oracle import preprocess
    fr = _structured_frames(2, 11)
    img, _ = det.debug(fr)
    ref = np.stack([preprocess(f) for f in fr])
    assert img.shape == (2, 256, 256, 3)
    # same weights, same float32 operation order: equal up to rounding ties of the uint8 quantisation (1 / 255)
    d = np.abs(img - ref)
    assert d.max() <= 1.0 / 255 + 1e-7 and (d > 1e-7).mean() < 1e-3
    assert img.min() >= 0.0 and img.max() <= 1.0


def test_network_and_decode_vs_oracle(det, state):
    from oracle.yolov4_oracle import YoloV4Oracle
    fr = _structured_frames(2, 21)
    img, maps = det.debug(fr)
    boxes, confs = det.forward(fr)
    o16 = YoloV4Oracle(state, "bf16")
    ref_maps = o16.raw_heads(img)                        # the oracle network on the SAME pre-processed image
    for m, r, hw in zip(maps, ref_maps, (32, 16, 8)):
        assert m.shape == (2, hw, hw, 255)
        rel = float(np.linalg.norm(m - r) / np.linalg.norm(r))
        print(f"detector map {hw}x{hw}: max|ref|={np.abs(r).max():.2f} max err={np.abs(m - r).max():.2e} rel L2={rel:.2e}")
        assert rel < 3e-2                                # bf16 re-rounding noise through up to 107 stored layers
    # decode kernel against the oracle's decode of the GPU's own maps: float32 transcendentals only
    b_ref, c_ref = YoloV4Oracle.decode(maps)
    assert boxes.shape == (2, 4032, 1, 4) and confs.shape == (2, 4032, 80)
    np.testing.assert_allclose(boxes, b_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(confs, c_ref, rtol=1e-5, atol=1e-6)
    # end to end against the oracle chain: the maps' bf16 re-rounding noise (a few % of their range, random synthetic
    # weights amplify it like any perturbation) through the sigmoids
    b_o, c_o = YoloV4Oracle.decode(ref_maps)
    dc = np.abs(confs - c_o)
    print(f"detector end to end: max|d conf|={dc.max():.2e} mean|d conf|={dc.mean():.2e} max|d box|={np.abs(boxes - b_o).max():.2e}")
    # conf = sigmoid(cls) * sigmoid(obj): sigmoid' <= 1/4 and both factors <= 1, so every confidence can move by at most a quarter
    # of the error of the two logits it is made of -- the map error (bounded above) carried through, element by element,
    # instead of a flat ceiling
    bound = []
    for m, r, hw in zip(maps, ref_maps, (32, 16, 8)):
        d = np.abs(m - r).reshape(2, hw * hw, 3, 85)                      # [b, cell, anchor, channel]
        bd = 0.25 * (d[..., 5:] + d[..., 4:5])                            # [b, cell, anchor, 80]
        bound.append(bd.transpose(0, 2, 1, 3).reshape(2, 3 * hw * hw, 80))   # box row = anchor * H * W + cell
    bound = np.concatenate(bound, axis=1)
    assert np.all(dc <= bound + 1e-5), float((dc - bound).max())
    assert dc.mean() < 2e-3
    # drift of the bf16 network against the pure-fp32 definition (informative)
    f32 = YoloV4Oracle(state, "f32").raw_heads(img)
    print("detector maps, bf16 HIP vs fp32 definition: rel L2 =",
          [round(float(np.linalg.norm(m - r) / np.linalg.norm(r)), 4) for m, r in zip(maps, f32)])


def test_device_path_batching_and_person_selection(det, state):
    import torch
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    fr = _structured_frames(6, 31)                        # max_batch = 4: two micro-batches
    boxes, confs = det.forward(fr)
    b2, c2 = det.forward(torch.from_numpy(fr).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(b2.cpu().numpy(), boxes) and np.array_equal(c2.cpu().numpy(), confs)
    one_b, one_c = det.forward(fr[4:5])                   # frames are independent
    assert np.array_equal(one_b[0], boxes[4]) and np.array_equal(one_c[0], confs[4])
    # the reference's post-processing on these tensors (hpe.py:60-79): GPU kernel == oracle, frame by frame
    eng = HpeEngine(device=0, max_batch=8)
    try:
        bbox, found = eng.select_person(boxes, confs, 0.3)
        bbox_d, found_d = eng.select_person(b2, c2, 0.3)                        # device tensors in, device tensors out
        torch.cuda.synchronize()
        assert np.array_equal(bbox_d.cpu().numpy(), bbox) and np.array_equal(found_d.cpu().numpy(), found)
        for i in range(6):
            sel = ho.select_person(boxes[i:i + 1], confs[i:i + 1], 640, 480)
            assert (sel is None and not found[i]) or tuple(bbox[i]) == sel
    finally:
        eng.close()


def test_large_micro_batches_are_bit_identical(det, state):
    """Micro-batches of up to 256 frames (1-GiB activation tensors, where the 32-bit-offset guards of the 3x3 kernels start to
    matter; +44 % frames/s over 64-frame micro-batches): frames are independent, so 160 frames in ONE micro-batch give the bits of
    the same frames four at a time."""
    from isbfsar_amd.det_engine import DetEngine
    fr = synth.frames(160, seed=910)
    big = DetEngine(device=0, max_batch=256)
    try:
        big.load_weights(state)
        b1, c1 = big.forward(fr)
    finally:
        big.close()
    b2, c2 = det.forward(fr)
    assert np.array_equal(b1, b2) and np.array_equal(c1, c2)
    assert np.isfinite(b1).all() and float(c1.max()) > 0


def test_concatenations_in_place_are_bit_identical(det, state, monkeypatch):
    """Convolutions whose output only a concatenation reads write straight into their channel slice of the joined tensor
    (ConvArgs.out_ld; both halves of the five CSP joins, one half of the four PANet joins): the same bits as with every
    concatenation copied (ISB_DET_ALIAS=0, read when the handle is created)."""
    from isbfsar_amd.det_engine import DetEngine
    fr = _structured_frames(5, 77)
    b1, c1 = det.forward(fr)
    monkeypatch.setenv("ISB_DET_ALIAS", "0")
    plain = DetEngine(device=0, max_batch=4)
    try:
        plain.load_weights(state)
        b0, c0 = plain.forward(fr)
    finally:
        plain.close()
    assert np.array_equal(b0, b1) and np.array_equal(c0, c1)


def test_estimator_uses_the_builtin_detector():
    """HumanPoseEstimator with detector weights configured: estimate() runs detector -> post-processing -> crop -> pose with
    no caller-supplied box (the reference's flow, hpe.py:51-173); just_box mode returns the detector's box."""
    from isbfsar_amd.modules.hpe.hpe import HumanPoseEstimator
    from isbfsar_amd.params import MetrabsHIPConfig, RealSenseIntrinsics
    cfg = MetrabsHIPConfig()
    cfg.yolo_synthetic = True
    cfg.yolo_thresh = 0.05           # synthetic detection weights: let some anchor through
    cfg.max_batch = 2
    est = HumanPoseEstimator(cfg, RealSenseIntrinsics(), just_box=True)
    assert est.det is not None
    frame = _structured_frames(1, 41)[0]
    r = est.estimate(frame)
    boxes, confs = est.det.forward(frame[None])
    person = (confs[0].argmax(-1) == 0) & (confs[0].max(-1) > 0.05)
    if person.any():
        assert r is not None and len(r["bbox"]) == 4 and all(isinstance(v, int) and v >= 0 for v in r["bbox"])
    else:
        assert r is None
    full = HumanPoseEstimator(cfg, RealSenseIntrinsics())
    out = full.estimate(frame)
    assert out is None or (out["pose"].shape == (30, 3) and np.isfinite(out["pose"]).all())
