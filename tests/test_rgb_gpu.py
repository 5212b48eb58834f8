"""GPU: the ResNet-50 trunk on the conv_igemm family (isb_rgb_*) against its CPU definition (parity unpinned: torchvision and
its weights are not in the reference tree), and the hybrid chain trunk -> PostResNet -> [rgb | sk] features -> transformer."""
import numpy as np
import pytest

from isbfsar_amd import resnet50, synth, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rgb():
    from isbfsar_amd.rgb_engine import RgbEngine
    e = RgbEngine(device=0, max_batch=4)
    st = resnet50.make_state(0)
    e.load_weights(st)
    yield e, st
    e.close()


def test_trunk_vs_oracle(rgb):
    from oracle.resnet50_oracle import ResNet50Oracle
    e, st = rgb
    x = np.random.default_rng(3).normal(0, 1, (6, 3, 224, 224)).astype(np.float32)       # 4 + 2: two micro-batches
    got = e.forward(x)
    ref = ResNet50Oracle(st, "bf16").forward(x)
    assert got.shape == (6, 2048) and np.isfinite(got).all()
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(f"ResNet-50 trunk: |ref| max {np.abs(ref).max():.3f}, rel L2 vs bf16-faithful oracle {rel:.2e}, max abs {np.abs(got - ref).max():.2e}")
    assert rel < 1e-2                                   # same storage points: accumulation order + bf16 re-rounding flips
    f32 = ResNet50Oracle(st, "f32").forward(x)
    assert np.linalg.norm(got - f32) / np.linalg.norm(f32) < 5e-2
    # layouts and entry points agree bit for bit
    import torch
    nhwc = np.ascontiguousarray(x.transpose(0, 2, 3, 1))
    assert np.array_equal(e.forward(nhwc, nchw=False), got)
    dev = e.forward(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), got)
    # images are independent
    assert np.array_equal(e.forward(x[2:3]), got[2:3])


def test_hybrid_chain_vs_oracle(rgb):
    """frames (person crops) + poses -> ResNet-50 trunk -> isb_ar_infer_hybrid, against the oracle chain fed with the SAME
    trunk features (the trunk itself is compared above; here: PostResNet, concatenation, 512-wide transformer, discriminator)."""
    from isbfsar_amd.engine import ArEngine
    from oracle.ar_oracle import TRXOSOracle
    e, _ = rgb
    L, J, way, B = 16, 30, 3, 2
    rng = np.random.default_rng(5)
    ss_img = rng.normal(0, 1, (way * L, 3, 224, 224)).astype(np.float32)
    q_img = rng.normal(0, 1, (B * L, 3, 224, 224)).astype(np.float32)
    ss_trunk = e.forward(ss_img).reshape(way, L, 2048)
    q_trunk = e.forward(q_img).reshape(B, L, 2048)
    state = weights.make_ar_state(L, J, seed=6, hybrid=True)
    ss = synth.skeleton_windows(way, L, J, seed=106)
    q = synth.skeleton_windows(B, L, J, seed=206)
    ar = ArEngine(L, J, way, device=0, precision="bf16x3", input_type="hybrid")
    ar.load_weights(state)
    ar.set_support(poses=ss, trunk=ss_trunk)
    logits, is_true, embed = ar.infer(q, want_embed=True, trunk=q_trunk)
    ref = TRXOSOracle(state, L, J, d_in=512).forward_hybrid(ss, ss_trunk, way, q, q_trunk)
    np.testing.assert_allclose(logits, ref["logits"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(is_true, ref["is_true"][:, 0], rtol=0, atol=1e-4)
    np.testing.assert_allclose(embed, ref["query_features"], rtol=0, atol=2e-5)


def test_action_recognizer_dropin_hybrid():
    """modules/ar/ar.py::ActionRecognizer with input_type "hybrid" (utils/params.py:81): frames {"rgb": [3,224,224], "sk": [3J]}
    (main.py:85-105), support classes {"imgs", "poses"} (ar.py:64-67), against the oracle chain on the trunk features the
    drop-in computed; cached support features and the warm-up contract as in skeleton mode."""
    from isbfsar_amd.modules.ar.ar import ActionRecognizer
    from isbfsar_amd.params import TRXConfig
    from oracle.ar_oracle import TRXOSOracle
    L, J, way = 16, 30, 5
    args = TRXConfig()
    args.input_type, args.seq_len, args.n_joints, args.way, args.device = "hybrid", L, J, way, "cuda"
    state = weights.make_ar_state(L, J, seed=8, hybrid=True)
    args.weights = state
    args.rgb_weights = resnet50.make_state(0)
    args.precision = "bf16x3"
    ar = ActionRecognizer(args)
    rng = np.random.default_rng(9)
    names = ["wave", "sit"]
    ss = synth.skeleton_windows(len(names), L, J, seed=108)
    imgs = rng.normal(0, 1, (len(names), L, 3, 224, 224)).astype(np.float32)
    for i, nm in enumerate(names):
        ar.train({"flag": nm, "data": {"poses": ss[i], "imgs": imgs[i]}, "requires_focus": False})
    stream = synth.skeleton_windows(1, L + 2, J, seed=208)[0]
    frames = rng.normal(0, 1, (L + 2, 3, 224, 224)).astype(np.float32)
    outs = []
    for t in range(L + 2):
        res = ar.inference({"rgb": frames[t], "sk": stream[t]})
        if t < L - 1:
            assert res == ({}, 0, {})
        else:
            outs.append(res)
    assert all("features" in v for v in ar.support_set.values())
    assert tuple(ar.support_set["wave"]["features"].shape) == (L, 512)
    net = TRXOSOracle(state, L, J, d_in=512)
    ss_trunk = ar.rgb.forward(imgs.reshape(-1, 3, 224, 224)).reshape(len(names), L, 2048)
    for k, (res, is_true, _) in enumerate(outs):
        t = L - 1 + k
        q_trunk = ar.rgb.forward(frames[t - L + 1:t + 1])[None]
        ref = net.forward_hybrid(ss, ss_trunk, len(names), stream[t - L + 1:t + 1][None], q_trunk)
        lg = ref["logits"][0]
        p = np.exp(lg - lg.max()); p /= p.sum()
        np.testing.assert_allclose([res[n] for n in names], p, rtol=0, atol=1e-4)
        np.testing.assert_allclose(is_true, ref["is_true"][0], rtol=0, atol=1e-4)
