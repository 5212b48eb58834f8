"""ResNet-50 trunk of the RGB / hybrid action-recognition branch: description, deterministic synthetic weights, checkpoint
key mapping.

The reference builds it as ``nn.Sequential(*list(torchvision.models.resnet50(pretrained=True).children())[:-1])``
(modules/ar/utils/model.py:270-277): conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2 -> layer1..4 (Bottleneck [3,4,6,3],
stride on the 3x3 conv = torchvision's "v1.5") -> global average pool, i.e. ``[N,3,224,224]`` (ImageNet-normalised, main.py:86-92)
-> ``[N,2048,1,1]``. torchvision and its pretrained weights are not in the reference tree ("parity unpinned", like the pose
backbone): the block table below is the PUBLIC torchvision architecture, checked in tests/test_rgb_cpu.py against its
published parameter count (23,508,032 trainable without the classifier) and 4.09 GMAC per 224 x 224 image.

Blob tensor names (all f32; conv weights [cout, kh, kw, cin] = K-contiguous rows, BatchNorm folded to scale/shift):
    rgb.conv1.{w,scale,shift}
    rgb.layer{1..4}.{i}.conv{1,2,3}.{w,scale,shift}
    rgb.layer{1..4}.0.downsample.{w,scale,shift}
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Mapping, Tuple

import numpy as np

from .weights import uniform

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))      # (planes, blocks, stride of the first block)
EXPANSION = 4
BN_EPS = 1e-5
FEAT = 2048
IMG = 224


@dataclass
class Bottleneck:
    layer: int          # 1..4
    idx: int            # block index inside the layer
    cin: int
    planes: int
    stride: int         # on conv2 (3x3): torchvision v1.5
    downsample: bool
    in_hw: int
    out_hw: int

    @property
    def cout(self) -> int:
        return self.planes * EXPANSION

    @property
    def prefix(self) -> str:
        return f"rgb.layer{self.layer}.{self.idx}"


def blocks(img: int = IMG) -> List[Bottleneck]:
    out: List[Bottleneck] = []
    hw = img // 4                       # conv1 /2, maxpool /2
    cin = 64
    for li, (planes, n, stride) in enumerate(LAYERS, start=1):
        for i in range(n):
            st = stride if i == 0 else 1
            out.append(Bottleneck(li, i, cin, planes, st, i == 0 and (st != 1 or cin != planes * EXPANSION), hw, hw // st))
            hw //= st
            cin = planes * EXPANSION
    return out


def tensor_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(prefix, cout, k, cin):
        s[prefix + ".w"] = (cout, k, k, cin)
        s[prefix + ".scale"] = (cout,)
        s[prefix + ".shift"] = (cout,)

    conv("rgb.conv1", 64, 7, 3)
    for b in blocks():
        conv(b.prefix + ".conv1", b.planes, 1, b.cin)
        conv(b.prefix + ".conv2", b.planes, 3, b.planes)
        conv(b.prefix + ".conv3", b.cout, 1, b.planes)
        if b.downsample:
            conv(b.prefix + ".downsample", b.cout, 1, b.cin)
    return s


def count_parameters() -> int:
    """trainable parameters of torchvision's resnet50 without its fc layer: conv kernels + BatchNorm gamma / beta"""
    n = 0
    for name, shape in tensor_shapes().items():
        if name.endswith(".w"):
            n += int(np.prod(shape)) + 2 * shape[0]
    return n


def macs_per_image(img: int = IMG) -> int:
    m = (img // 2) ** 2 * 64 * 49 * 3
    for b in blocks(img):
        m += b.in_hw ** 2 * b.cin * b.planes                        # conv1 1x1 (at the input resolution)
        m += b.out_hw ** 2 * 9 * b.planes * b.planes                # conv2 3x3 (strided)
        m += b.out_hw ** 2 * b.planes * b.cout                      # conv3 1x1
        if b.downsample:
            m += b.out_hw ** 2 * b.cin * b.cout
    return m


def make_state(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic weights that keep activations O(1) through the 16 blocks: He-style gain on the activated
    convs, a small gain on each block's last conv (its output is added to the skip path), folded-BN scale in [0.8, 1.2]."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in tensor_shapes().items():
        leaf = name.split(".")[-1]
        if leaf == "scale":
            out[name] = uniform(name, shape, 0.8, 1.2, seed)
        elif leaf == "shift":
            out[name] = uniform(name, shape, -0.05, 0.05, seed)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.5 if ".conv3." in name else (1.0 if ".downsample." in name else 1.4)
            a = gain * np.sqrt(3.0 / fan_in)
            out[name] = uniform(name, shape, -a, a, seed)
    return out


# ------------------------------------------------------------------------------------------------
# Real-weight ingestion: the keys of a hybrid checkpoint's RGB trunk. Inside TRXOS the trunk is
# an nn.Sequential of resnet50's children minus fc (model.py:274), so its state_dict keys are `features_extractor.rgb.<i>...`
# with i = 0 conv1, 1 bn1, 4..7 layer1..4 (2 relu, 3 maxpool, 8 avgpool carry no tensors). A plain torchvision state_dict
# (`conv1.weight`, `layer1.0.conv1.weight`, ...) is accepted too. Layout contract only: no torchvision / checkpoint here to
# verify against (exercised by a round trip in tests/test_rgb_cpu.py).
# ------------------------------------------------------------------------------------------------
_SEQ = {"0": "conv1", "1": "bn1", "4": "layer1", "5": "layer2", "6": "layer3", "7": "layer4"}


def _fold(sd, conv_key, bn_key):
    w = np.asarray(sd[conv_key + ".weight"], np.float32)                       # OIHW
    g, b = np.asarray(sd[bn_key + ".weight"], np.float32), np.asarray(sd[bn_key + ".bias"], np.float32)
    mu, var = np.asarray(sd[bn_key + ".running_mean"], np.float32), np.asarray(sd[bn_key + ".running_var"], np.float32)
    scale = g / np.sqrt(var + np.float32(BN_EPS))
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)), scale.astype(np.float32), (b - mu * scale).astype(np.float32)


def state_from_torch(state_dict: Mapping[str, "object"], prefix: str = "features_extractor.rgb.") -> "OrderedDict[str, np.ndarray]":
    sd = {}
    for k, v in state_dict.items():
        k = k.replace(".module", "")
        if prefix and k.startswith(prefix):
            head, _, rest = k[len(prefix):].partition(".")
            k = _SEQ.get(head, head) + ("." + rest if rest else "")
        sd[k] = np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def put(name, conv_key, bn_key):
        out[name + ".w"], out[name + ".scale"], out[name + ".shift"] = _fold(sd, conv_key, bn_key)

    put("rgb.conv1", "conv1", "bn1")
    for b in blocks():
        t = f"layer{b.layer}.{b.idx}"
        for i in (1, 2, 3):
            put(f"{b.prefix}.conv{i}", f"{t}.conv{i}", f"{t}.bn{i}")
        if b.downsample:
            put(f"{b.prefix}.downsample", f"{t}.downsample.0", f"{t}.downsample.1")
    for name, shape in tensor_shapes().items():
        if out[name].shape != shape:
            raise ValueError(f"{name}: shape {out[name].shape}, expected {shape}")
    return out


def to_torch_state(state: Mapping[str, np.ndarray], prefix: str = "features_extractor.rgb.") -> "OrderedDict[str, np.ndarray]":
    """inverse of state_from_torch (BatchNorm with mean 0, variance 1 - eps): the round-trip check of the key mapping"""
    inv = {v: k for k, v in _SEQ.items()}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def key(k):
        head, _, rest = k.partition(".")
        return prefix + inv[head] + ("." + rest if rest else "") if prefix else k

    def put(name, conv_key, bn_key):
        out[key(conv_key + ".weight")] = np.ascontiguousarray(state[name + ".w"].transpose(0, 3, 1, 2))
        out[key(bn_key + ".weight")] = state[name + ".scale"].copy()
        out[key(bn_key + ".bias")] = state[name + ".shift"].copy()
        out[key(bn_key + ".running_mean")] = np.zeros_like(state[name + ".scale"])
        out[key(bn_key + ".running_var")] = np.full_like(state[name + ".scale"], 1.0 - BN_EPS)

    put("rgb.conv1", "conv1", "bn1")
    for b in blocks():
        t = f"layer{b.layer}.{b.idx}"
        for i in (1, 2, 3):
            put(f"{b.prefix}.conv{i}", f"{t}.conv{i}", f"{t}.bn{i}")
        if b.downsample:
            put(f"{b.prefix}.downsample", f"{t}.downsample.0", f"{t}.downsample.1")
    return out
