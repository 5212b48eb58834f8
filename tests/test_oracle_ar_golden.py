"""CPU: the numpy restatement (oracle/ar_oracle.py) against vectors captured from the
reference's own TRXOS (oracle/gen_golden.py). Pins the oracle."""
import hashlib
import os

import numpy as np
import pytest

from isbfsar_amd import synth, weights
from oracle.ar_oracle import ActionRecognizerOracle, TRXOSOracle


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _state_kw(g):
    """the weight-generator switches a fixture was made with (oracle/gen_golden.py: the "sharp" fixtures)"""
    return {k: float(g[k]) for k in ("disc_gain", "norm_gain") if k in g.files}


@pytest.mark.parametrize("name", ["ar_ref_16_30_5.npz", "ar_bl_30_122_60.npz", "ar_bl_30_122_120.npz",
                                  "ar_sharp_16_30_5.npz", "ar_sharp_30_122_60.npz"])
def test_trxos_oracle_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    L, J, way, B, seed = (int(g[k]) for k in ("L", "J", "way", "B", "seed"))
    state = weights.make_ar_state(L, J, seed=seed, **_state_kw(g))
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100)
    q = synth.skeleton_windows(B, L, J, seed=seed + 200)
    # the synthetic generator must reproduce the inputs the reference saw
    assert _digest(ss) == str(g["ss_digest"]) and _digest(q) == str(g["q_digest"])
    net = TRXOSOracle(state, L, J)
    out = net.forward(ss, way, q)
    sharp = "sharp" in name        # LayerNorm gain x3: logits up to -14; discriminator x6: f32 noise amplified 6^4 times
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=0, atol=2e-4 if sharp else 2e-5)
    np.testing.assert_allclose(out["is_true"], g["is_true"], rtol=0, atol=1e-4 if sharp else 2e-6)
    if sharp:                      # the point of these fixtures: the open-set score moves
        assert g["is_true"].min() < 0.1 and g["is_true"].max() > 0.9
    np.testing.assert_allclose(out["query_features"][0], g["qfeat0"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["support_features"][0], g["support_features_c0"], rtol=0, atol=2e-6)
    assert abs(float(out["support_features"].astype(np.float64).sum()) - float(g["support_features_sum"])) < 1e-2
    if "kq0" in g:
        kq, vq = net.tuples_kv(out["query_features"][:1])
        np.testing.assert_allclose(kq[0], g["kq0"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(vq[0], g["vq0"], rtol=0, atol=2e-6)
    # G3: cached support features give the same answer (ar.py:56-61)
    out2 = net.forward(None, way, q, ss_features=out["support_features"])
    assert np.array_equal(out2["logits"], out["logits"])


def test_trxos_oracle_float64_agrees(golden_dir):
    g = _load(golden_dir, "ar_ref_16_30_5.npz")
    L, J, way, B, seed = (int(g[k]) for k in ("L", "J", "way", "B", "seed"))
    net = TRXOSOracle(weights.make_ar_state(L, J, seed=seed), L, J, dtype=np.float64)
    out = net.forward(g["ss"], way, g["q"])
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["is_true"], g["is_true"], rtol=0, atol=2e-6)


def test_sliding_window_state_machine(golden_dir):
    g = _load(golden_dir, "ar_stream_ref_16_30_5.npz")
    L, J, way, seed, n_frames = (int(g[k]) for k in ("L", "J", "way", "seed", "n_frames"))
    net = TRXOSOracle(weights.make_ar_state(L, J, seed=seed), L, J)
    ar = ActionRecognizerOracle(net, way)
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100)
    stream = synth.skeleton_windows(1, n_frames, J, seed=seed + 300)[0]
    assert _digest(stream) == str(g["stream_digest"])
    assert ar.inference({"sk": stream[0]}) == ({}, 0, {})          # no classes yet (ar.py:37-38)
    for c in range(way):
        ar.train({"flag": f"c{c}", "data": {"poses": ss[c]}, "requires_focus": False})
    k = 0
    for t in range(n_frames):
        res, is_true, rf = ar.inference({"sk": stream[t]})
        if t < L - 1:
            assert (res, is_true, rf) == ({}, 0, {})               # warm-up (ar.py:43-44)
            continue
        p = np.array([res[f"c{c}"] for c in range(way)])
        np.testing.assert_allclose(p, g["probs"][k], rtol=0, atol=1e-5)
        np.testing.assert_allclose(is_true, g["is_true"][k], rtol=0, atol=2e-6)
        assert all("features" in v for v in ar.support_set.values())  # cached after first call (ar.py:72-74)
        k += 1
    assert k == len(g["probs"])
    assert ar.remove("c0") and not ar.remove("c0")


def _ckpt_dialects(g):
    """the reference-produced state_dict of tests/golden/ar_ckpt_*.npz under each of its three key dialects"""
    import json
    tensors = {k[3:]: g[k] for k in g.files if k.startswith("t::")}
    return {name: {new: tensors[plain] for plain, new in keymap.items()} for name, keymap in json.loads(str(g["dialects"])).items()}


@pytest.mark.parametrize("dialect", ["plain", "dataparallel", "pre_rgb"])
def test_converter_on_reference_checkpoint(golden_dir, dialect):
    """weights.state_from_torch on a state_dict the REFERENCE's TRXOS produced (torch default init,
    oracle/gen_golden.py::gen_ar_checkpoint) in the three dialects a DISC.pth comes in (ar.py:17-19,
    rename_torch_layers_and_parameters.py:9-13): the converted weights, run through the oracle, give the outputs that
    very TRXOS computed."""
    g = _load(golden_dir, "ar_ckpt_ref_16_30_5.npz")
    sd = _ckpt_dialects(g)[dialect]
    if dialect == "dataparallel":
        assert all(".module." in k for k in sd)
    if dialect == "pre_rgb":
        assert not any(k.startswith("post_resnet.") or ".sk." in k for k in sd)
    L, J, way = (int(g[k]) for k in ("L", "J", "way"))
    state = weights.state_from_torch(sd, keys=set(weights.ar_state_shapes(L, J)))
    assert set(state) == set(weights.ar_state_shapes(L, J))           # pe buffer and post_resnet dropped, nothing missing
    for k, shape in weights.ar_state_shapes(L, J).items():
        assert state[k].shape == shape and state[k].dtype == np.float32
    out = TRXOSOracle(state, L, J).forward(g["ss"], way, g["q"])
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["is_true"], g["is_true"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["query_features"], g["qfeat"], rtol=0, atol=2e-6)


def test_trxos_oracle_hybrid_matches_reference(golden_dir):
    """input_type "hybrid" (utils/params.py:81, model.py:270-277, 296-316): PostResNet on the RGB trunk's features, the
    [rgb | sk] concatenation, 512-wide transformer input -- against the reference's TRXOS run with a stand-in trunk
    (oracle/gen_golden.py::gen_ar_hybrid); the fixture carries the trunk outputs."""
    g = _load(golden_dir, "ar_hybrid_16_30_5.npz")
    L, J, way, B, seed = (int(g[k]) for k in ("L", "J", "way", "B", "seed"))
    state = weights.make_ar_state(L, J, seed=seed, hybrid=True)
    assert state["transformers.0.k_linear.weight"].shape == (128, 1024) and state["post_resnet.l1.weight"].shape == (256, 2048)
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100)
    q = synth.skeleton_windows(B, L, J, seed=seed + 200)
    assert _digest(ss) == str(g["ss_digest"]) and _digest(q) == str(g["q_digest"])
    net = TRXOSOracle(state, L, J, d_in=512)
    out = net.forward_hybrid(ss, g["ss_trunk"], way, q, g["q_trunk"])
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["is_true"], g["is_true"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["support_features"], g["support_features"], rtol=0, atol=2e-6)
    out2 = net.forward_hybrid(None, None, way, q, g["q_trunk"], ss_features=out["support_features"])
    assert np.array_equal(out2["logits"], out["logits"])
