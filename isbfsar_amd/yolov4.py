"""YOLOv4 person detector: layer table, deterministic synthetic weights, checkpoint conversion.

The reference runs the detector as an opaque TensorRT engine (``yolo.engine``: reference modules/hpe/hpe.py:42,59-60)
exported from the un-vendored Tianxiaomo/pytorch-YOLOv4 ``Yolov4(n_classes=80, inference=True)`` at 256 x 256
(modules/hpe/setup/1_extract_yolo_onnx.py:4-12,21-25,44-60). Neither its definition nor its weights are in the reference
tree ("parity unpinned"); what the tree pins is the CONTRACT: input ``f32[1,3,256,256]`` RGB in [0,1] made
from the BGR frame by ``cv2.resize(..., INTER_AREA)`` (hpe.py:51-56), outputs ``boxes [1,4032,1,4]`` (x1,y1,x2,y2
normalised) and ``confs [1,4032,80]`` (hpe.py:60), 4032 = 3 anchors x (32^2 + 16^2 + 8^2), class 0 = person (hpe.py:67).
The table below is the public YOLOv4 architecture (CSPDarknet53 + SPP + PANet + three YOLO heads, Mish in the backbone,
LeakyReLU(0.1) in neck and head, BatchNorm eps 1e-5), with the module names of that implementation so that a real
``yolov4.pth`` converts by key (``state_from_torch``).

Blob tensor names (f32; conv weights ``[cout, k, k, cin]``; BatchNorm folded to scale / shift, the three detection convs
have scale 1 and shift = their bias):  ``yolo.<module path>.{w,scale,shift}``, e.g. ``yolo.down1.conv1.w``.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .weights import uniform

ACT_LINEAR, ACT_MISH, ACT_LEAKY = 0, 2, 3          # activation codes of the conv kernels (1 = SiLU, the pose backbone's)
N_CLASSES = 80
IN_HW = 256
ANCHORS = (12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401)
# per detection scale: stride, anchor indices, scale_x_y (Yolov4Head in the public implementation)
SCALES = ((8, (0, 1, 2), 1.2), (16, (3, 4, 5), 1.1), (32, (6, 7, 8), 1.05))
N_BOXES = sum(3 * (IN_HW // s) ** 2 for s, _, _ in SCALES)   # 4032
BN_EPS = 1e-5


@dataclass
class Conv:
    name: str
    cin: int
    cout: int
    k: int
    stride: int
    act: int
    bn: bool = True     # False: the detection convs (bias, no BatchNorm, no activation)


def conv_layers() -> List[Conv]:
    """Every convolution in module order (the order the C++ plan builder walks, isb_det_describe)."""
    L: List[Conv] = []

    def c(name, cin, cout, k, stride=1, act=ACT_MISH, bn=True):
        L.append(Conv(name, cin, cout, k, stride, act, bn))

    def resblock(prefix, ch, n):
        for i in range(n):
            c(f"{prefix}.module_list.{i}.0", ch, ch, 1)
            c(f"{prefix}.module_list.{i}.1", ch, ch, 3)

    # CSPDarknet53
    c("down1.conv1", 3, 32, 3); c("down1.conv2", 32, 64, 3, 2); c("down1.conv3", 64, 64, 1); c("down1.conv4", 64, 64, 1)
    c("down1.conv5", 64, 32, 1); c("down1.conv6", 32, 64, 3); c("down1.conv7", 64, 64, 1); c("down1.conv8", 128, 64, 1)
    for d, ch, n in ((2, 64, 2), (3, 128, 8), (4, 256, 8), (5, 512, 4)):
        p = f"down{d}"
        c(f"{p}.conv1", ch, 2 * ch, 3, 2); c(f"{p}.conv2", 2 * ch, ch, 1); c(f"{p}.conv3", 2 * ch, ch, 1)
        resblock(f"{p}.resblock", ch, n)
        c(f"{p}.conv4", ch, ch, 1); c(f"{p}.conv5", 2 * ch, 2 * ch, 1)
    # SPP + PANet neck (LeakyReLU)
    lk = ACT_LEAKY
    for name, cin, cout, k in (("conv1", 1024, 512, 1), ("conv2", 512, 1024, 3), ("conv3", 1024, 512, 1), ("conv4", 2048, 512, 1),
                               ("conv5", 512, 1024, 3), ("conv6", 1024, 512, 1), ("conv7", 512, 256, 1), ("conv8", 512, 256, 1),
                               ("conv9", 512, 256, 1), ("conv10", 256, 512, 3), ("conv11", 512, 256, 1), ("conv12", 256, 512, 3),
                               ("conv13", 512, 256, 1), ("conv14", 256, 128, 1), ("conv15", 256, 128, 1), ("conv16", 256, 128, 1),
                               ("conv17", 128, 256, 3), ("conv18", 256, 128, 1), ("conv19", 128, 256, 3), ("conv20", 256, 128, 1)):
        c(f"neek.{name}", cin, cout, k, 1, lk)
    # heads
    no = 3 * (5 + N_CLASSES)       # 255
    c("head.conv1", 128, 256, 3, 1, lk); c("head.conv2", 256, no, 1, 1, ACT_LINEAR, False)
    c("head.conv3", 128, 256, 3, 2, lk)
    for name, cin, cout, k in (("conv4", 512, 256, 1), ("conv5", 256, 512, 3), ("conv6", 512, 256, 1), ("conv7", 256, 512, 3),
                               ("conv8", 512, 256, 1), ("conv9", 256, 512, 3)):
        c(f"head.{name}", cin, cout, k, 1, lk)
    c("head.conv10", 512, no, 1, 1, ACT_LINEAR, False)
    c("head.conv11", 256, 512, 3, 2, lk)
    for name, cin, cout, k in (("conv12", 1024, 512, 1), ("conv13", 512, 1024, 3), ("conv14", 1024, 512, 1), ("conv15", 512, 1024, 3),
                               ("conv16", 1024, 512, 1), ("conv17", 512, 1024, 3)):
        c(f"head.{name}", cin, cout, k, 1, lk)
    c("head.conv18", 1024, no, 1, 1, ACT_LINEAR, False)
    return L


def tensor_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for l in conv_layers():
        s[f"yolo.{l.name}.w"] = (l.cout, l.k, l.k, l.cin)
        s[f"yolo.{l.name}.scale"] = (l.cout,)
        s[f"yolo.{l.name}.shift"] = (l.cout,)
    return s


def macs_per_frame() -> int:
    """multiply-accumulates of one 256 x 256 frame (the spatial size of every conv follows from the strides)"""
    hw = {}
    size = IN_HW
    m = 0
    # sizes: down1 256 -> 128 after conv2; each downN.conv1 halves; neck / head sizes by position
    sizes = {"down1.conv1": 256}
    for l in conv_layers():
        n = l.name
        if n == "down1.conv1":
            o = 256
        elif n.startswith("down1"):
            o = 128
        elif n.startswith("down2"):
            o = 64
        elif n.startswith("down3"):
            o = 32
        elif n.startswith("down4"):
            o = 16
        elif n.startswith("down5"):
            o = 8
        elif n.startswith("neek"):
            i = int(n.split("conv")[1])
            o = 8 if i <= 7 else (16 if i <= 14 else 32)
        else:
            i = int(n.split("conv")[1])
            o = 32 if i <= 2 else (16 if i <= 10 else 8)
        m += o * o * l.k * l.k * l.cin * l.cout
    return m


def make_state(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic weights: He-style gain for the activated convs (Mish / LeakyReLU keep about half the
    variance), folded-BN scale in [0.8,1.2], shift in [-0.05,0.05]; the detection convs get a gain that spreads the
    objectness / class logits over a few units so that some anchors clear the 0.3 threshold and most do not."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for l in conv_layers():
        p = f"yolo.{l.name}"
        fan_in = l.k * l.k * l.cin
        gain = 1.0 if not l.bn else (1.45 if l.act != ACT_LINEAR else 1.0)
        if ".module_list." in l.name and l.name.endswith(".1"):
            gain = 0.6                      # the residual branch of a ResBlock: keep the sum's variance near constant
        a = gain * np.sqrt(3.0 / fan_in)
        out[p + ".w"] = uniform(p + ".w", (l.cout, l.k, l.k, l.cin), -a, a, seed)
        if l.bn:
            out[p + ".scale"] = uniform(p + ".scale", (l.cout,), 0.8, 1.2, seed)
            out[p + ".shift"] = uniform(p + ".shift", (l.cout,), -0.05, 0.05, seed)
        else:
            out[p + ".scale"] = np.ones((l.cout,), np.float32)
            out[p + ".shift"] = uniform(p + ".shift", (l.cout,), -2.5, -0.5, seed)     # detection bias: few confident anchors
    return out


def state_from_torch(state_dict, eps: float = BN_EPS) -> "OrderedDict[str, np.ndarray]":
    """A ``Yolov4`` state dict of the public implementation (``yolov4.pth`` after the reference's 'neek' -> 'neck' rewrite,
    1_extract_yolo_onnx.py:30-40, or before it) -> this repo's blob tensors. Conv_Bn_Activation keeps its layers in
    ``<module>.conv.0`` (Conv2d, OIHW) and ``<module>.conv.1`` (BatchNorm2d); the three detection convs have a bias and no
    BatchNorm. Layout contract only: there is no checkpoint in the tree to verify it on."""
    sd = {k.replace("neck.", "neek."): np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float32)
          for k, v in state_dict.items()}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for l in conv_layers():
        p = f"yolo.{l.name}"
        w = sd[f"{l.name}.conv.0.weight"]                                   # [O,I,kh,kw]
        out[p + ".w"] = np.ascontiguousarray(np.transpose(w, (0, 2, 3, 1)))  # -> [O,kh,kw,I]
        if l.bn:
            g, b = sd[f"{l.name}.conv.1.weight"], sd[f"{l.name}.conv.1.bias"]
            mu, var = sd[f"{l.name}.conv.1.running_mean"], sd[f"{l.name}.conv.1.running_var"]
            sc = g / np.sqrt(var + eps)
            out[p + ".scale"] = sc.astype(np.float32)
            out[p + ".shift"] = (b - mu * sc).astype(np.float32)
        else:
            out[p + ".scale"] = np.ones((l.cout,), np.float32)
            out[p + ".shift"] = sd[f"{l.name}.conv.0.bias"]
    shapes = tensor_shapes()
    for k, a in out.items():
        if tuple(a.shape) != shapes[k]:
            raise ValueError(f"{k}: converted shape {a.shape} != expected {shapes[k]}")
    return out
