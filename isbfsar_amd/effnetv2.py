"""EfficientNetV2-L backbone description + deterministic synthetic weights.

The reference names the backbone only by string: ``model_name = 'efficientnetv2-l'``,
``get_model(..., include_top=False)`` from the un-vendored isarandi/metrabs repo
(reference modules/hpe/setup/2_extract_bbone_heads.py:27,46-47), shipped as a TensorRT engine
taking ``f32[B,256,256,3]`` NHWC and returning ``f32[B,8,8,1280]`` (7_create_engines.py:38-42,
4_create_heads_onnx.py:13,19). Neither source nor weights are in the reference tree, so the block
table below is the PUBLIC efficientnetv2-l configuration ("parity unpinned"):

    stem  conv3x3/s2 3->32, BN, SiLU
    r=4  FusedMBConv e1 k3 s1  32-> 32      r=7  FusedMBConv e4 k3 s2  32-> 64
    r=7  FusedMBConv e4 k3 s2  64-> 96      r=10 MBConv e4 k3 s2 se.25  96->192
    r=19 MBConv e6 k3 s1 se.25 192->224     r=25 MBConv e6 k3 s2 se.25 224->384
    r=7  MBConv e6 k3 s1 se.25 384->640     head conv1x1 640->1280, BN, SiLU
    BN eps 1e-3 (folded to per-channel scale/shift), TF "SAME" padding (stride 2: pad bottom/right
    only), SE squeeze = max(1, int(block_in * 0.25)), residual when stride 1 and in == out.

The same table exists in C++ (csrc/hpe_api.cpp, kStages) -- isb_hpe_load_weights checks every blob tensor's
shape against it, and tests/test_oracle_effnetv2.py checks this table against the published block strings.

Blob tensor names (all f32; conv weights are [cout, kh, kw, cin], i.e. K-contiguous rows):
    bbone.stem.{w,scale,shift}
    bbone.b{i}.expand.{w,scale,shift}     3x3 (fused) or 1x1 (MBConv); for e1 fused blocks this is the only conv
    bbone.b{i}.dw.{w,scale,shift}         [c,3,3] depthwise (MBConv)
    bbone.b{i}.se.{w1,b1,w2,b2}           [cse,c],[cse],[c,cse],[c]
    bbone.b{i}.project.{w,scale,shift}    1x1
    bbone.head.{w,scale,shift}
    head.{weight,bias}                    MetrABS heatmap head Linear(1280,288) (4_create_heads_onnx.py:10)
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .weights import uniform

# (kind, repeats, expand, stride, cin, cout, se)
STAGES = [
    ("fused", 4, 1, 1, 32, 32, 0.0),
    ("fused", 7, 4, 2, 32, 64, 0.0),
    ("fused", 7, 4, 2, 64, 96, 0.0),
    ("mb", 10, 4, 2, 96, 192, 0.25),
    ("mb", 19, 6, 1, 192, 224, 0.25),
    ("mb", 25, 6, 2, 224, 384, 0.25),
    ("mb", 7, 6, 1, 384, 640, 0.25),
]
STEM_OUT = 32
HEAD_OUT = 1280
N_HEAD_LOGITS = 288


@dataclass
class Block:
    idx: int
    kind: str       # "fused" | "mb"
    cin: int
    cout: int
    cexp: int
    stride: int
    cse: int        # 0 = no SE
    residual: bool
    in_hw: int      # input spatial size for a 256x256 crop
    out_hw: int


def blocks(in_hw: int = 128) -> List[Block]:
    out: List[Block] = []
    hw = in_hw
    i = 0
    for kind, rep, e, s, cin, cout, se in STAGES:
        for r in range(rep):
            bi = cin if r == 0 else cout
            st = s if r == 0 else 1
            ohw = hw // st
            out.append(Block(i, kind, bi, cout, bi * e, st, max(1, int(bi * se)) if se > 0 else 0,
                             st == 1 and bi == cout, hw, ohw))
            hw = ohw
            i += 1
    return out


def macs_per_crop() -> int:
    """MACs of the backbone + pose head for one 256x256 crop (15.99 G)."""
    m = 128 * 128 * 27 * STEM_OUT
    for b in blocks():
        o = b.out_hw * b.out_hw
        if b.kind == "fused":
            m += o * 9 * b.cin * b.cexp
            if b.cexp != b.cin:
                m += o * b.cexp * b.cout
        else:
            m += b.in_hw * b.in_hw * b.cin * b.cexp + o * 9 * b.cexp + o * b.cexp * b.cout + 2 * b.cexp * b.cse
    m += 64 * 640 * HEAD_OUT + 64 * HEAD_OUT * N_HEAD_LOGITS
    return m


def tensor_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(prefix, cout, k, cin):
        s[prefix + ".w"] = (cout, k, k, cin)
        s[prefix + ".scale"] = (cout,)
        s[prefix + ".shift"] = (cout,)

    conv("bbone.stem", STEM_OUT, 3, 3)
    for b in blocks():
        p = f"bbone.b{b.idx}"
        if b.kind == "fused":
            if b.cexp == b.cin:                      # expand ratio 1: a single 3x3 conv
                conv(p + ".expand", b.cout, 3, b.cin)
            else:
                conv(p + ".expand", b.cexp, 3, b.cin)
                conv(p + ".project", b.cout, 1, b.cexp)
        else:
            conv(p + ".expand", b.cexp, 1, b.cin)
            s[p + ".dw.w"] = (b.cexp, 3, 3)
            s[p + ".dw.scale"] = (b.cexp,)
            s[p + ".dw.shift"] = (b.cexp,)
            s[p + ".se.w1"] = (b.cse, b.cexp)
            s[p + ".se.b1"] = (b.cse,)
            s[p + ".se.w2"] = (b.cexp, b.cse)
            s[p + ".se.b2"] = (b.cexp,)
            conv(p + ".project", b.cout, 1, b.cexp)
    conv("bbone.head", HEAD_OUT, 1, 640)
    s["head.weight"] = (N_HEAD_LOGITS, HEAD_OUT)
    s["head.bias"] = (N_HEAD_LOGITS,)
    return s


def make_state(seed: int = 0, profile: str = "default", head_gain: float = 4.0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic weights that keep activations O(1) through all 79 blocks:
    activated convs get He-style gain, projections (no activation, added to the residual)
    a small gain, folded-BN scale in [0.8,1.2] and shift in [-0.1,0.1].

    profile "default": every projection has gain 0.35. The six stage transitions (no identity skip) then shrink
        the input-dependent part of the activations by 0.35 each while the BN shifts keep feeding O(0.1) in: the final
        features barely depend on the frame (frame-to-frame difference 0.3 % of their magnitude) and the pose heatmaps
        are flat. Fine for throughput, weak as a parity input.
    profile "signal": projections of the transition blocks have gain 1.0, those of the residual blocks 0.1, shifts in
        [-0.02, 0.02]: the activations keep a standard deviation of 1-3.5 through all stages, the final features
        (std 1.1, max 10) differ between frames by as much as their own magnitude and the heatmaps are PEAKED (the
        regime of a trained MetrABS). The parity tests of the pose stage use this profile."""
    if profile not in ("default", "signal"):
        raise ValueError(f"unknown weight profile {profile!r}")
    sig = profile == "signal"
    bl = {b.idx: b for b in blocks()}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in tensor_shapes().items():
        leaf = name.split(".")[-1]
        if leaf == "scale":
            out[name] = uniform(name, shape, 0.8, 1.2, seed)
        elif leaf in ("shift", "b1", "b2", "bias"):
            sh = 0.02 if sig else 0.1
            out[name] = uniform(name, shape, -sh, sh, seed)
        else:
            fan_in = int(np.prod(shape[1:]))
            if ".project." in name:
                gain = 0.35
                if sig:
                    gain = 0.1 if bl[int(name.split(".")[1][1:])].residual else 1.0
            elif name == "head.weight":
                gain = head_gain           # spread the pose-head logits so heatmaps are not flat
            elif ".se.w2" in name:
                gain = 1.0
            else:
                gain = 1.6
            a = gain * np.sqrt(3.0 / fan_in)
            out[name] = uniform(name, shape, -a, a, seed)
    return out


# ------------------------------------------------------------------------------------------------
# Real-weight ingestion. The reference extracts the backbone from a MetrABS SavedModel
# (modules/hpe/setup/2_extract_bbone_heads.py:35-60) whose EfficientNetV2-L lives in the un-vendored
# isarandi/metrabs repo, so there is no checkpoint here to test against: what follows is the layout contract
# (TF HWIO kernels + BatchNorm statistics -> this repo's OHWI weights + folded scale/shift), with the variable
# names of the public efficientnetv2 implementation the reference names ('efficientnetv2-l', line 27). It is
# exercised by a round trip through `to_keras_variables` (tests/test_abi.py); it is NOT verified on a real export.
# ------------------------------------------------------------------------------------------------
BN_EPS = 1e-3


def _bn_suffix(i: int) -> str:
    return "tpu_batch_normalization" if i == 0 else f"tpu_batch_normalization_{i}"


def _conv_suffix(i: int) -> str:
    return "conv2d" if i == 0 else f"conv2d_{i}"


def keras_name_map() -> "OrderedDict[str, dict]":
    """blob conv prefix -> {'kernel': tf name, 'bn': tf BN scope} (+ SE / depthwise entries) for every layer."""
    m: "OrderedDict[str, dict]" = OrderedDict()
    m["bbone.stem"] = {"kernel": "stem/conv2d/kernel", "bn": "stem/tpu_batch_normalization"}
    for b in blocks():
        scope = f"blocks_{b.idx}"
        p = f"bbone.b{b.idx}"
        if b.kind == "fused":
            m[p + ".expand"] = {"kernel": f"{scope}/conv2d/kernel", "bn": f"{scope}/{_bn_suffix(0)}"}
            if b.cexp != b.cin:
                m[p + ".project"] = {"kernel": f"{scope}/conv2d_1/kernel", "bn": f"{scope}/{_bn_suffix(1)}"}
        else:
            m[p + ".expand"] = {"kernel": f"{scope}/conv2d/kernel", "bn": f"{scope}/{_bn_suffix(0)}"}
            m[p + ".dw"] = {"kernel": f"{scope}/depthwise_conv2d/depthwise_kernel", "bn": f"{scope}/{_bn_suffix(1)}"}
            m[p + ".se"] = {"reduce": f"{scope}/se/conv2d", "expand": f"{scope}/se/conv2d_1"}
            m[p + ".project"] = {"kernel": f"{scope}/conv2d_1/kernel", "bn": f"{scope}/{_bn_suffix(2)}"}
    m["bbone.head"] = {"kernel": "head/conv2d/kernel", "bn": "head/tpu_batch_normalization"}
    return m


def state_from_keras(variables, head_kernel=None, head_bias=None, eps: float = BN_EPS) -> "OrderedDict[str, np.ndarray]":
    """TF/Keras variables (name -> array, names as in `keras_name_map`) -> this repo's state:
    conv kernels HWIO -> [O,H,W,I]; depthwise [3,3,C,1] -> [C,3,3]; SE 1x1 convs [1,1,I,O] -> [O,I];
    BatchNorm folded: scale = gamma / sqrt(moving_variance + eps), shift = beta - moving_mean * scale.
    `head_kernel` [1,1,1280,288] / `head_bias` = MetrABS `heatmap_heads.conv_final`
    (2_extract_bbone_heads.py:66-67, 4_create_heads_onnx.py:22-25: weight = W.squeeze().T)."""
    v = {k.split(":")[0]: np.asarray(a, dtype=np.float32) for k, a in variables.items()}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def bn(scope):
        g, b2 = v[scope + "/gamma"], v[scope + "/beta"]
        mu, var = v[scope + "/moving_mean"], v[scope + "/moving_variance"]
        sc = g / np.sqrt(var + eps)
        return sc.astype(np.float32), (b2 - mu * sc).astype(np.float32)

    for p, e in keras_name_map().items():
        if p.endswith(".se"):
            r, x = e["reduce"], e["expand"]
            out[p + ".w1"] = np.ascontiguousarray(v[r + "/kernel"][0, 0].T)          # [cse, C]
            out[p + ".b1"] = v[r + "/bias"]
            out[p + ".w2"] = np.ascontiguousarray(v[x + "/kernel"][0, 0].T)          # [C, cse]
            out[p + ".b2"] = v[x + "/bias"]
            continue
        k = v[e["kernel"]]
        if p.endswith(".dw"):
            out[p + ".w"] = np.ascontiguousarray(np.transpose(k[:, :, :, 0], (2, 0, 1)))   # [C,3,3]
        else:
            out[p + ".w"] = np.ascontiguousarray(np.transpose(k, (3, 0, 1, 2)))            # HWIO -> OHWI
        out[p + ".scale"], out[p + ".shift"] = bn(e["bn"])
    if head_kernel is not None:
        out["head.weight"] = np.ascontiguousarray(np.asarray(head_kernel, np.float32).squeeze().T)
        out["head.bias"] = np.asarray(head_bias, np.float32)
    shapes = tensor_shapes()
    for name, a in out.items():
        if tuple(a.shape) != tuple(shapes[name]):
            raise ValueError(f"{name}: converted shape {a.shape} != expected {shapes[name]}")
    # same tensor order as tensor_shapes()
    return OrderedDict((n, out[n]) for n in shapes if n in out)


def to_keras_variables(state, eps: float = BN_EPS):
    """Inverse of `state_from_keras` with gamma = scale, beta = shift, mean 0, variance 1 - eps (test helper)."""
    v = {}
    for p, e in keras_name_map().items():
        if p.endswith(".se"):
            v[e["reduce"] + "/kernel"] = np.ascontiguousarray(state[p + ".w1"].T)[None, None]
            v[e["reduce"] + "/bias"] = state[p + ".b1"]
            v[e["expand"] + "/kernel"] = np.ascontiguousarray(state[p + ".w2"].T)[None, None]
            v[e["expand"] + "/bias"] = state[p + ".b2"]
            continue
        w = state[p + ".w"]
        if p.endswith(".dw"):
            v[e["kernel"]] = np.ascontiguousarray(np.transpose(w, (1, 2, 0)))[..., None]
        else:
            v[e["kernel"]] = np.ascontiguousarray(np.transpose(w, (1, 2, 3, 0)))
        c = w.shape[0]
        v[e["bn"] + "/gamma"] = state[p + ".scale"]
        v[e["bn"] + "/beta"] = state[p + ".shift"]
        v[e["bn"] + "/moving_mean"] = np.zeros(c, np.float32)
        v[e["bn"] + "/moving_variance"] = np.full(c, 1.0 - eps, np.float32)
    return v
