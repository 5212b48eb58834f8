"""ORACLE (test infrastructure, NOT product code) -- PARITY UNPINNED.

CPU definition (PyTorch fp32 ops) of the YOLOv4 person detector the reference runs as an opaque TensorRT engine
(``yolo.engine``: reference modules/hpe/hpe.py:42,59-60), exported from the un-vendored Tianxiaomo/pytorch-YOLOv4
``Yolov4(n_classes=80, inference=True)`` at 256 x 256 with downloaded weights
(modules/hpe/setup/1_extract_yolo_onnx.py:4-25,44-60). Neither definition nor weights are in the reference tree, so
there is no reference output to pin against; this file restates the PUBLIC architecture (CSPDarknet53 + SPP + PANet +
three YOLO heads and their box decoding) on this repo's blob tensors and the HIP detector is compared with it on
synthetic weights. Pinned by the reference tree: the input contract (hpe.py:51-56) and the output contract
``boxes [B,4032,1,4]``, ``confs [B,4032,80]`` (hpe.py:60); the post-processing that consumes them is pinned separately
(oracle/hpe_oracle.py, reference misc.py:27-107).

Modes as in oracle/effnetv2_oracle.py: ``"f32"`` plain fp32; ``"bf16"`` the storage points of the HIP path (conv weights
with the folded BatchNorm scale and every stored activation rounded to bf16, accumulation / bias / activation in fp32,
the first conv on the fp32 image with fp32 weights, the three detection convs stored in fp32).
Only tests/ and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np
import torch
import torch.nn.functional as F

ANCHORS = (12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401)
SCALES = ((8, (0, 1, 2), 1.2), (16, (3, 4, 5), 1.1), (32, (6, 7, 8), 1.05))
N_CLASSES = 80


def area_resize_u8(frame: np.ndarray, out_hw: int = 256) -> np.ndarray:
    """cv2.resize(frame, (256, 256), interpolation=INTER_AREA) for a uint8 image (hpe.py:51): every output pixel is the
    area-weighted mean of the source rectangle it covers, rounded to the nearest integer (ties to even). Separable, rows
    first then columns, float32 products and sums in ascending source order (the order the HIP kernel uses)."""
    H, W, _ = frame.shape

    def taps(n_in, n_out):
        sc = n_in / n_out
        out = []
        for o in range(n_out):
            lo, hi = o * sc, (o + 1) * sc
            idx = list(range(int(np.floor(lo)), min(int(np.ceil(hi)), n_in)))
            out.append([(i, np.float32(max(0.0, min(hi, i + 1) - max(lo, i)) / sc)) for i in idx])
        return out

    img = frame.astype(np.float32)
    rows = np.zeros((out_hw, W, 3), np.float32)
    for o, tp in enumerate(taps(H, out_hw)):
        acc = np.zeros((W, 3), np.float32)
        for i, w in tp:
            acc = acc + w * img[i]
        rows[o] = acc
    out = np.zeros((out_hw, out_hw, 3), np.float32)
    for o, tp in enumerate(taps(W, out_hw)):
        acc = np.zeros((out_hw, 3), np.float32)
        for i, w in tp:
            acc = acc + w * rows[:, i]
        out[:, o] = acc
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def preprocess(frame_bgr: np.ndarray) -> np.ndarray:
    """hpe.py:51-56: area resize to 256 x 256, BGR -> RGB, / 255. Returns f32 [256,256,3] (NHWC; the reference's NCHW is
    the same numbers)."""
    sq = area_resize_u8(frame_bgr)
    return (sq[..., ::-1].astype(np.float32) / np.float32(255.0)).copy()


def _r(x: torch.Tensor, mode: str) -> torch.Tensor:
    return x.bfloat16().float() if mode == "bf16" else x


def _mish(x):
    return x * torch.tanh(F.softplus(x))


class YoloV4Oracle:
    def __init__(self, state: Mapping[str, np.ndarray], mode: str = "bf16"):
        assert mode in ("f32", "bf16")
        self.mode = mode
        self.w: Dict[str, torch.Tensor] = {}
        self.shift: Dict[str, torch.Tensor] = {}
        for k, v in state.items():
            if k.endswith(".w"):
                p = k[:-2]
                w = torch.from_numpy(np.ascontiguousarray(v, np.float32)) * torch.from_numpy(
                    np.ascontiguousarray(state[p + ".scale"], np.float32)).view(-1, 1, 1, 1)
                if p != "yolo.down1.conv1":
                    w = _r(w, mode)
                self.w[p] = w.permute(0, 3, 1, 2).contiguous()            # [O,kh,kw,I] -> OIHW
                self.shift[p] = torch.from_numpy(np.ascontiguousarray(state[p + ".shift"], np.float32))

    def conv(self, x, name, act, stride=1, keep_f32=False):
        p = "yolo." + name
        w = self.w[p]
        k = w.shape[-1]
        y = F.conv2d(x, w, stride=stride, padding=(k - 1) // 2) + self.shift[p].view(1, -1, 1, 1)
        if act == "mish":
            y = _mish(y)
        elif act == "leaky":
            y = F.leaky_relu(y, 0.1)
        return y if keep_f32 else _r(y, self.mode)

    def _csp(self, x, p, ch, n):
        c = self.conv
        x1 = c(x, f"{p}.conv1", "mish", 2)
        x2 = c(x1, f"{p}.conv2", "mish")
        x3 = c(x1, f"{p}.conv3", "mish")
        for i in range(n):
            h = c(c(x3, f"{p}.resblock.module_list.{i}.0", "mish"), f"{p}.resblock.module_list.{i}.1", "mish", keep_f32=True)
            x3 = _r(x3 + h, self.mode)
        x4 = c(x3, f"{p}.conv4", "mish")
        return c(torch.cat([x4, x2], 1), f"{p}.conv5", "mish")

    def raw_heads(self, images_nhwc: np.ndarray):
        """images f32 [B,256,256,3] RGB in [0,1] -> the three detection maps f32 [B,H,W,255] (strides 8, 16, 32)."""
        c = self.conv
        x = torch.from_numpy(np.ascontiguousarray(images_nhwc, np.float32)).permute(0, 3, 1, 2)
        with torch.no_grad():
            x1 = c(x, "down1.conv1", "mish")
            x2 = c(x1, "down1.conv2", "mish", 2)
            x3 = c(x2, "down1.conv3", "mish")
            x4 = c(x2, "down1.conv4", "mish")
            x6 = c(c(x4, "down1.conv5", "mish"), "down1.conv6", "mish", keep_f32=True)
            x6 = _r(x6 + x4, self.mode)
            x7 = c(x6, "down1.conv7", "mish")
            d1 = c(torch.cat([x7, x3], 1), "down1.conv8", "mish")
            d2 = self._csp(d1, "down2", 64, 2)
            d3 = self._csp(d2, "down3", 128, 8)
            d4 = self._csp(d3, "down4", 256, 8)
            d5 = self._csp(d4, "down5", 512, 4)
            n3 = c(c(c(d5, "neek.conv1", "leaky"), "neek.conv2", "leaky"), "neek.conv3", "leaky")
            m1, m2, m3 = (F.max_pool2d(n3, k, 1, k // 2) for k in (5, 9, 13))
            n6 = c(c(c(torch.cat([m3, m2, m1, n3], 1), "neek.conv4", "leaky"), "neek.conv5", "leaky"), "neek.conv6", "leaky")
            up = F.interpolate(c(n6, "neek.conv7", "leaky"), scale_factor=2, mode="nearest")
            n = torch.cat([c(d4, "neek.conv8", "leaky"), up], 1)
            for i in range(9, 14):
                n = c(n, f"neek.conv{i}", "leaky")
            n13 = n
            up = F.interpolate(c(n13, "neek.conv14", "leaky"), scale_factor=2, mode="nearest")
            n = torch.cat([c(d3, "neek.conv15", "leaky"), up], 1)
            for i in range(16, 21):
                n = c(n, f"neek.conv{i}", "leaky")
            n20 = n
            o1 = c(c(n20, "head.conv1", "leaky"), "head.conv2", None, keep_f32=True)
            h = torch.cat([c(n20, "head.conv3", "leaky", 2), n13], 1)
            for i in range(4, 9):
                h = c(h, f"head.conv{i}", "leaky")
            h8 = h
            o2 = c(c(h8, "head.conv9", "leaky"), "head.conv10", None, keep_f32=True)
            h = torch.cat([c(h8, "head.conv11", "leaky", 2), n6], 1)
            for i in range(12, 18):
                h = c(h, f"head.conv{i}", "leaky")
            o3 = c(h, "head.conv18", None, keep_f32=True)
        return [o.permute(0, 2, 3, 1).contiguous().numpy() for o in (o1, o2, o3)]

    @staticmethod
    def decode(maps):
        """YoloLayer inference of the public implementation (yolo_forward_dynamic): per scale and anchor a,
        bxy = sigmoid(t) * s - (s - 1) / 2 + grid, bwh = exp(t) * anchor / stride, all divided by the grid size;
        boxes (x1, y1, x2, y2); confs = sigmoid(cls) * sigmoid(obj). Box index = a * H * W + y * W + x, scales in
        the order stride 8, 16, 32. Returns boxes [B,4032,1,4], confs [B,4032,80] (float32)."""
        boxes, confs = [], []
        for (stride, mask, sxy), o in zip(SCALES, maps):
            B, H, W, _ = o.shape
            o = o.astype(np.float32).reshape(B, H, W, 3, 5 + N_CLASSES).transpose(0, 3, 1, 2, 4)      # [B,A,H,W,85]
            sig = lambda v: (1.0 / (1.0 + np.exp(-v.astype(np.float64)))).astype(np.float32)
            gx = np.arange(W, dtype=np.float32)[None, None, None, :]
            gy = np.arange(H, dtype=np.float32)[None, None, :, None]
            aw = np.array([ANCHORS[2 * a] / stride for a in mask], np.float32)[None, :, None, None]
            ah = np.array([ANCHORS[2 * a + 1] / stride for a in mask], np.float32)[None, :, None, None]
            bx = (sig(o[..., 0]) * np.float32(sxy) - np.float32(0.5 * (sxy - 1)) + gx) / np.float32(W)
            by = (sig(o[..., 1]) * np.float32(sxy) - np.float32(0.5 * (sxy - 1)) + gy) / np.float32(H)
            bw = np.exp(o[..., 2]) * aw / np.float32(W)
            bh = np.exp(o[..., 3]) * ah / np.float32(H)
            x1, y1 = bx - bw * np.float32(0.5), by - bh * np.float32(0.5)
            bb = np.stack([x1, y1, x1 + bw, y1 + bh], -1).reshape(B, 3 * H * W, 1, 4)
            cf = (sig(o[..., 5:]) * sig(o[..., 4:5])).reshape(B, 3 * H * W, N_CLASSES)
            boxes.append(bb.astype(np.float32))
            confs.append(cf.astype(np.float32))
        return np.concatenate(boxes, 1), np.concatenate(confs, 1)

    def forward(self, images_nhwc: np.ndarray):
        return self.decode(self.raw_heads(images_nhwc))
