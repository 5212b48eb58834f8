"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

Imports the reference's own classes (never copies their source), loads OUR deterministic
weights (``isbfsar_amd.weights``) into them, runs them on OUR seeded synthetic inputs
(``isbfsar_amd.synth``) and stores inputs' digests + outputs as small ``.npz`` fixtures under
``tests/golden``. The fixtures are data; this script is what made them.

    python oracle/gen_golden.py            # regenerates every fixture

Reference entry points exercised:
    modules/ar/utils/model.py:219  TRXOS            (torchvision stubbed: only the RGB branch uses it)
    modules/hpe/utils/misc.py:243  homography       (imported unmodified)          [hpe goldens]
    modules/hpe/hpe.py:48          HumanPoseEstimator.estimate through a fake Runner [hpe goldens]
"""
from __future__ import annotations

import hashlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from isbfsar_amd import synth, weights  # noqa: E402


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _stub_torchvision():
    import torch

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tvm.resnet50 = None
    tvm.ResNet = torch.nn.Module            # only subclassed at class-definition time (model.py:221)
    tvr.Bottleneck = None
    tvr.resnet18 = None
    tv.models = tvm
    tvm.resnet = tvr
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr})


def reference_trxos(seq_len: int, n_joints: int, way: int, state):
    """Instantiate the reference's TRXOS on CPU with our weights."""
    import torch

    _stub_torchvision()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from modules.ar.utils.model import TRXOS
        from utils.params import TRXConfig
    finally:
        os.chdir(cwd)
    args = TRXConfig()
    args.device = "cpu"
    args.seq_len, args.n_joints, args.way = seq_len, n_joints, way
    net = TRXOS(args).eval()
    sd = net.state_dict()
    new = {}
    for k, v in sd.items():
        if k in state:
            assert tuple(v.shape) == state[k].shape, (k, v.shape, state[k].shape)
            new[k] = torch.from_numpy(np.array(state[k]))
        else:
            new[k] = v                         # pe buffer, unused post_resnet
    net.load_state_dict(new)
    return net


def gen_ar(tag: str, L: int, J: int, way: int, B: int, seed: int, keep_intermediates: bool, **state_kw):
    import torch

    # state_kw: disc_gain / norm_gain (weights.make_ar_state) -- the "sharp" fixtures, whose open-set score spans
    # (0.05, 0.95) and whose LayerNorm gain is that of a trained model
    state = weights.make_ar_state(L, J, seed=seed, **state_kw)
    net = reference_trxos(L, J, way, state)
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100)
    q = synth.skeleton_windows(B, L, J, seed=seed + 200)
    labels = torch.arange(way, dtype=torch.int32)[None]
    outs = {"logits": [], "is_true": [], "qfeat": []}
    with torch.no_grad():
        # the reference's batch dimension carries its own copy of the support set (model.py:59-63)
        for b in range(B):
            o = net({"sk": torch.from_numpy(ss)[None]}, labels, {"sk": torch.from_numpy(q[b:b + 1])})
            outs["logits"].append(o["logits"].numpy()[0])
            outs["is_true"].append(o["is_true"].numpy()[0])
            sf = o["support_features"].numpy()[0]
            if b == 0:
                # G3: cached-feature call (ar.py:56-61) equals the raw-pose call
                o2 = net(None, labels, {"sk": torch.from_numpy(q[b:b + 1])}, ss_features=o["support_features"])
                assert np.array_equal(o2["logits"].numpy(), o["logits"].numpy())
                assert np.array_equal(o2["is_true"].numpy(), o["is_true"].numpy())
                qf = net.features_extractor["sk"](torch.from_numpy(q[b:b + 1])).numpy()[0]
                outs["qfeat"] = qf
                tr = net.transformers[0]
                if keep_intermediates:
                    # Kq/Vq of query 0 as the reference computes them (model.py:65-84)
                    x = tr.pe(torch.from_numpy(qf)[None, None])
                    tq = torch.stack([torch.index_select(x, -2, p).reshape(1, 1, -1) for p in tr.tuples], dim=-2)
                    outs["kq0"] = tr.norm_k(tr.k_linear(tq)).numpy()[0, 0]
                    outs["vq0"] = tr.v_linear(tq).numpy()[0, 0]
                    outs["proto0_c0"] = o["prototypes"][0].numpy()[0, 0]
    rec = dict(
        L=L, J=J, way=way, B=B, seed=seed,
        disc_gain=np.float64(state_kw.get("disc_gain", 1.0)), norm_gain=np.float64(state_kw.get("norm_gain", 1.0)),
        ss_digest=digest(ss), q_digest=digest(q),
        logits=np.stack(outs["logits"]), is_true=np.stack(outs["is_true"]),
        qfeat0=outs["qfeat"], support_features_c0=sf[0], support_features_digest=digest(sf),
        support_features_sum=np.float64(sf.astype(np.float64).sum()),
    )
    for k in ("kq0", "vq0", "proto0_c0"):
        if k in outs:
            rec[k] = outs[k]
    if L * J * way <= 16 * 30 * 5:
        rec["ss"] = ss
        rec["q"] = q
    np.savez_compressed(os.path.join(OUT, f"ar_{tag}.npz"), **rec)
    print(f"ar_{tag}: logits[0,:5]={rec['logits'][0, :5]} is_true={rec['is_true'].ravel()[:4]}")


def gen_ar_stream(tag: str, L: int, J: int, way: int, n_frames: int, seed: int):
    """G7: the per-call outputs ActionRecognizer.inference (ar.py:30-84) produces for a frame
    stream; the wrapper itself needs CUDA + an absent checkpoint (ar.py:17,20), so TRXOS is driven
    with exactly the window contents the wrapper would assemble (ar.py:42-50)."""
    import torch

    state = weights.make_ar_state(L, J, seed=seed)
    net = reference_trxos(L, J, way, state)
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100)
    stream = synth.skeleton_windows(1, n_frames, J, seed=seed + 300)[0]   # [n_frames, 3J]
    labels = torch.arange(way, dtype=torch.int32)[None]
    probs, is_true = [], []
    with torch.no_grad():
        for t in range(L - 1, n_frames):
            win = torch.from_numpy(stream[t - L + 1:t + 1])[None]
            o = net({"sk": torch.from_numpy(ss)[None]}, labels, {"sk": win})
            probs.append(torch.softmax(o["logits"].squeeze(0), dim=0).numpy())   # ar.py:77
            is_true.append(o["is_true"].squeeze(0).numpy())                      # ar.py:78
    np.savez_compressed(os.path.join(OUT, f"ar_stream_{tag}.npz"), L=L, J=J, way=way, seed=seed,
                        n_frames=n_frames, stream_digest=digest(stream), ss_digest=digest(ss),
                        probs=np.stack(probs), is_true=np.stack(is_true))
    print(f"ar_stream_{tag}: {len(probs)} calls, probs[0]={probs[0]}")


def gen_ar_hybrid(tag: str, L: int, J: int, way: int, B: int, seed: int):
    """SURVEY 8f row 4 tail -- the HYBRID input type (TRXConfig.input_type = "hybrid", utils/params.py:81: transformer input 512):
    the reference's TRXOS with `{"rgb", "sk"}` inputs (model.py:270-277, 296-316). torchvision is not installed, so
    `resnet50(pretrained=True)` (model.py:275) is a stand-in trunk of the same interface -- children = [1x1 conv 3 -> 2048,
    global average pool, fc]; TRXOS drops the last child exactly as it does for the real network -- which pins everything
    AFTER the trunk: PostResNet (ReLU + Linear 2048 -> 256), the [rgb | sk] concatenation order, the 512-wide positional
    encoding and tuple Linear, the discriminator. The ResNet-50 itself stays unpinned (oracle/resnet50_oracle.py). The
    fixture stores the trunk's outputs (the [.., L, 2048] tensors the HIP path takes from its own ResNet-50 engine)."""
    import torch

    _stub_torchvision()

    class StandInTrunk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 2048, 1, bias=False)
            self.avgpool = torch.nn.AdaptiveAvgPool2d(1)
            self.fc = torch.nn.Linear(2048, 10)

    sys.modules["torchvision.models"].resnet50 = lambda pretrained=True: StandInTrunk()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import modules.ar.utils.model as ref_model
        from utils.params import TRXConfig
    finally:
        os.chdir(cwd)
    ref_model.resnet50 = sys.modules["torchvision.models"].resnet50      # model.py:6 bound the name at import time
    args = TRXConfig()
    args.device = "cpu"
    args.input_type = "hybrid"
    args.trans_linear_in_dim = 512                                      # utils/params.py:81
    args.seq_len, args.n_joints, args.way = L, J, way
    torch.manual_seed(seed)
    net = ref_model.TRXOS(args).eval()
    state = weights.make_ar_state(L, J, seed=seed, hybrid=True)
    sd = net.state_dict()
    new = {k: (torch.from_numpy(np.array(state[k])) if k in state else v) for k, v in sd.items()}
    for k in state:
        assert k in sd and tuple(sd[k].shape) == state[k].shape, k
    net.load_state_dict(new)
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100)
    q = synth.skeleton_windows(B, L, J, seed=seed + 200)
    rng = np.random.default_rng(seed)
    ss_img = rng.normal(0, 1, (way, L, 3, 8, 8)).astype(np.float32)
    q_img = rng.normal(0, 1, (B, L, 3, 8, 8)).astype(np.float32)
    labels = torch.arange(way, dtype=torch.int32)[None]
    with torch.no_grad():
        trunk = net.features_extractor["rgb"]
        ss_trunk = trunk(torch.from_numpy(ss_img).reshape(-1, 3, 8, 8)).reshape(way, L, -1).numpy()
        q_trunk = trunk(torch.from_numpy(q_img).reshape(-1, 3, 8, 8)).reshape(B, L, -1).numpy()
        logits, is_true, qfeat = [], [], []
        for b in range(B):
            o = net({"rgb": torch.from_numpy(ss_img)[None], "sk": torch.from_numpy(ss)[None]}, labels,
                    {"rgb": torch.from_numpy(q_img[b:b + 1]), "sk": torch.from_numpy(q[b:b + 1])})
            logits.append(o["logits"].numpy()[0])
            is_true.append(o["is_true"].numpy()[0])
            sf = o["support_features"].numpy()[0]
            if b == 0:      # cached-feature call (ar.py:56-61) equals the raw call
                o2 = net(None, labels, {"rgb": torch.from_numpy(q_img[b:b + 1]), "sk": torch.from_numpy(q[b:b + 1])},
                         ss_features=o["support_features"])
                assert np.array_equal(o2["logits"].numpy(), o["logits"].numpy())
    np.savez_compressed(os.path.join(OUT, f"ar_hybrid_{tag}.npz"), L=L, J=J, way=way, B=B, seed=seed,
                        ss_trunk=ss_trunk, q_trunk=q_trunk,
                        ss_digest=digest(ss), q_digest=digest(q), logits=np.stack(logits), is_true=np.stack(is_true),
                        support_features=sf)
    print(f"ar_hybrid_{tag}: logits[0]={logits[0]} is_true={np.stack(is_true).ravel()} trunk |max|={np.abs(q_trunk).max():.3f}")


def gen_ar_checkpoint(tag: str, L: int, J: int, way: int, B: int, torch_seed: int):
    """SURVEY 8f row 3 -- pins the checkpoint converter (isbfsar_amd/weights.state_from_torch). The reference's TRXOS is
    instantiated with torch's DEFAULT initialisation (seeded), i.e. a state_dict the reference itself produced, in the
    shape a DISC.pth['model_state_dict'] has (modules/ar/ar.py:17-19): every tensor of it is stored together with the
    key names of the three dialects such a checkpoint comes in, and with the outputs the same TRXOS computes:
      plain         the module's own keys (features_extractor.sk.*, transformers.0.*, discriminator.*, post_resnet.l1.*,
                    the transformers.0.pe.pe buffer)
      dataparallel  '.module' infixes as left by DataParallel wrappers, stripped by ar.py:18
      pre_rgb       checkpoints written before the RGB branch: features_extractor.fc1/fc2 (no '.sk' level), no
                    post_resnet.* -- what utils/rename_torch_layers_and_parameters.py:9-13 migrates (it adds '.sk' and
                    zero-filled post_resnet tensors; post_resnet is stored zero-filled here too: it is not on the
                    skeleton path and 2 MB of random numbers would be dead weight in the fixture)"""
    import json
    import torch

    _stub_torchvision()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from modules.ar.utils.model import TRXOS
        from utils.params import TRXConfig
    finally:
        os.chdir(cwd)
    args = TRXConfig()
    args.device = "cpu"
    args.seq_len, args.n_joints, args.way = L, J, way
    torch.manual_seed(torch_seed)
    net = TRXOS(args).eval()
    with torch.no_grad():                       # rename_torch_layers_and_parameters.py:12-13
        net.post_resnet.l1.weight.zero_()
        net.post_resnet.l1.bias.zero_()
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    plain = list(sd)
    dialects = {
        "plain": {k: k for k in plain},
        "dataparallel": {k: k.split(".", 1)[0] + ".module." + k.split(".", 1)[1] for k in plain},
        "pre_rgb": {k: k.replace("features_extractor.sk.", "features_extractor.") for k in plain if not k.startswith("post_resnet.")},
    }
    ss = synth.skeleton_windows(way, L, J, seed=torch_seed + 100)
    q = synth.skeleton_windows(B, L, J, seed=torch_seed + 200)
    labels = torch.arange(way, dtype=torch.int32)[None]
    logits, is_true = [], []
    with torch.no_grad():
        for b in range(B):
            o = net({"sk": torch.from_numpy(ss)[None]}, labels, {"sk": torch.from_numpy(q[b:b + 1])})
            logits.append(o["logits"].numpy()[0])
            is_true.append(o["is_true"].numpy()[0])
        qfeat = net.features_extractor["sk"](torch.from_numpy(q)).numpy()
    rec = {"t::" + k: v for k, v in sd.items()}
    rec.update(L=L, J=J, way=way, B=B, torch_seed=torch_seed, torch_version=str(torch.__version__),
               dialects=json.dumps(dialects), ss=ss, q=q, logits=np.stack(logits), is_true=np.stack(is_true), qfeat=qfeat)
    np.savez_compressed(os.path.join(OUT, f"ar_ckpt_{tag}.npz"), **rec)
    print(f"ar_ckpt_{tag}: {len(sd)} tensors, {sum(v.size for v in sd.values())} values, logits[0]={rec['logits'][0]} is_true={rec['is_true'].ravel()}")


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["ar", "hpe", "ckpt", "hybrid"]
    if "ar" in which:
        gen_ar("ref_16_30_5", 16, 30, 5, B=3, seed=0, keep_intermediates=True)
        gen_ar("bl_30_122_60", 30, 122, 60, B=2, seed=1, keep_intermediates=False)
        gen_ar_stream("ref_16_30_5", 16, 30, 5, n_frames=20, seed=0)
        # BASELINE configs[4]: 120-class support set (one window is enough: the reference takes 0.3 s per window here)
        gen_ar("bl_30_122_120", 30, 122, 120, B=1, seed=1, keep_intermediates=False)
        # resolving power: discriminator weights x6 (is_true spans 0.0-1.0 instead of 0.50-0.51) and LayerNorm gain x3
        gen_ar("sharp_16_30_5", 16, 30, 5, B=8, seed=2, keep_intermediates=False, disc_gain=6.0, norm_gain=3.0)
        gen_ar("sharp_30_122_60", 30, 122, 60, B=4, seed=2, keep_intermediates=False, disc_gain=6.0, norm_gain=3.0)
    if "hybrid" in which:
        gen_ar_hybrid("16_30_5", 16, 30, 5, B=3, seed=4)
    if "ckpt" in which or not sys.argv[1:]:
        gen_ar_checkpoint("ref_16_30_5", 16, 30, 5, B=4, torch_seed=1234)
    if "hpe" in which:
        try:
            from gen_golden_hpe import gen_all
        except ImportError:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from gen_golden_hpe import gen_all
        gen_all(OUT)


if __name__ == "__main__":
    main()
