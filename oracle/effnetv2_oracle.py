"""ORACLE (test infrastructure, NOT product code) -- PARITY UNPINNED.

CPU definition (PyTorch fp32 ops) of the EfficientNetV2-L backbone the reference runs as an opaque
TensorRT engine (``bbone1.engine``: reference utils/params.py:29, modules/hpe/hpe.py:103; contract
``f32[B,256,256,3]`` NHWC in [0,1] -> ``f32[B,8,8,1280]``: modules/hpe/setup/7_create_engines.py:38-42,
4_create_heads_onnx.py:13,19; model name 'efficientnetv2-l', include_top=False:
2_extract_bbone_heads.py:27,46-47). The arithmetic lives in isarandi/metrabs (un-vendored, no pinned
version) + downloaded weights, so there is NO reference output to pin against: this file restates
the PUBLIC efficientnetv2-l definition from its published block strings (``V2_L_BLOCKS`` below, parsed here --
nothing is imported from the product package) and the HIP backbone is compared with it on synthetic
weights. What IS pinned: the shape contract, and three known answers of the public model that
tests/test_oracle_effnetv2.py asserts on this file's own table: 117,746,848 parameters without the
classifier top (117,234,272 trainable: the figures Keras prints for EfficientNetV2L(include_top=False)),
15.99 GMAC per 256x256 crop and the output shape [B,8,8,1280].

Numeric modes:
  * ``mode="f32"``   plain fp32 everywhere;
  * ``mode="f16"``   the storage/rounding points of the HIP path's DEFAULT precision (isb_hpe_cfg.precision = 0 / 2; the
    precision the reference's TensorRT engines are built with, 7_create_engines.py:10): conv weights incl. the depthwise
    taps (BN scale folded in) and every stored activation -- stem output included -- rounded to IEEE fp16 (saturating);
    accumulation / bias / SiLU / SE in fp32, the last 1x1 conv stores f32 and the pose head runs in f32;
  * ``mode="bf16"``  isb_hpe_cfg.precision = 3 (round 3's layout): conv
    weights incl. the depthwise taps (BN scale folded in) and every stored activation are rounded to bf16 in the stem and
    in block strings 0-4; in the two 8x8 stages (block strings 5 and 6) and in the 640->1280 conv they are rounded to IEEE
    fp16 instead (the block that ENTERS the fp16 stages still runs its expand conv and its depthwise taps on bf16: its
    input is the bf16 stream); accumulation / bias / SiLU / SE in fp32, the last 1x1 conv stores f32 and the pose head runs
    in f32 -- so a GPU-vs-oracle difference is accumulation order only;
  * ``mode="bf16_plain"``  isb_hpe_cfg.precision = 1: bf16 at every rounding point (the round-2 layout).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, List, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

# the published efficientnetv2-l architecture strings: r = repeats, k = kernel, s = stride of the first repeat,
# e = expansion ratio, i / o = input / output filters, c1 = Fused-MBConv, se = squeeze-excite ratio of the block INPUT
V2_L_BLOCKS = (
    "r4_k3_s1_e1_i32_o32_c1",
    "r7_k3_s2_e4_i32_o64_c1",
    "r7_k3_s2_e4_i64_o96_c1",
    "r10_k3_s2_e4_i96_o192_se0.25",
    "r19_k3_s1_e6_i192_o224_se0.25",
    "r25_k3_s2_e6_i224_o384_se0.25",
    "r7_k3_s1_e6_i384_o640_se0.25",
)
STEM_FILTERS = 32          # = input filters of the first block
HEAD_FILTERS = 1280        # feature_size of the public config
BN_EPS = 1e-3


@dataclass
class OBlock:
    idx: int
    kind: str          # "fused" | "mb"
    cin: int
    cout: int
    cexp: int
    stride: int
    cse: int           # squeeze width, 0 = no SE
    residual: bool
    in_hw: int
    out_hw: int
    stage: int = 0     # index of the block string (0..6) the block was expanded from


def parse_block_string(s: str) -> dict:
    """'r10_k3_s2_e4_i96_o192_se0.25' -> dict(r=10, k=3, s=2, e=4, i=96, o=192, se=0.25, c=0)"""
    out = {"se": 0.0, "c": 0}
    for op in s.split("_"):
        m = re.match(r"^([a-z]+)([0-9.]+)$", op)
        if not m:
            raise ValueError(f"bad block option {op!r} in {s!r}")
        key, val = m.group(1), m.group(2)
        out[key] = float(val) if key == "se" else int(val)
    for need in ("r", "k", "s", "e", "i", "o"):
        if need not in out:
            raise ValueError(f"{s!r}: option {need!r} missing")
    return out


def oracle_blocks(in_hw: int = 128, strings=V2_L_BLOCKS) -> List[OBlock]:
    """Expand the block strings the way the public model builder does: the first repeat of a stage takes the stage's
    input filters and stride, later repeats run output->output at stride 1; squeeze width = max(1, int(block_in * se));
    identity skip when stride 1 and in == out."""
    out: List[OBlock] = []
    hw, idx = in_hw, 0
    for si, s in enumerate(strings):
        a = parse_block_string(s)
        assert a["k"] == 3, "efficientnetv2-l uses 3x3 kernels only"
        for rep in range(a["r"]):
            cin = a["i"] if rep == 0 else a["o"]
            stride = a["s"] if rep == 0 else 1
            ohw = hw // stride
            out.append(OBlock(idx, "fused" if a["c"] == 1 else "mb", cin, a["o"], cin * a["e"], stride,
                              max(1, int(cin * a["se"])) if a["se"] > 0 else 0, stride == 1 and cin == a["o"], hw, ohw, si))
            hw, idx = ohw, idx + 1
    return out


def count_parameters(blocks: Optional[List[OBlock]] = None) -> Dict[str, int]:
    """Parameters of the public model without its classifier top: convolution kernels (no bias), BatchNorm
    gamma/beta (trainable) + moving mean/variance, squeeze-excite kernels and biases."""
    blocks = oracle_blocks() if blocks is None else blocks
    w = STEM_FILTERS * 27
    bn = STEM_FILTERS
    for b in blocks:
        if b.kind == "fused":
            if b.cexp == b.cin:
                w += b.cout * 9 * b.cin
                bn += b.cout
            else:
                w += b.cexp * 9 * b.cin + b.cout * b.cexp
                bn += b.cexp + b.cout
        else:
            w += b.cexp * b.cin + 9 * b.cexp + b.cout * b.cexp + (b.cse * b.cexp + b.cse) + (b.cexp * b.cse + b.cexp)
            bn += 2 * b.cexp + b.cout
    w += HEAD_FILTERS * blocks[-1].cout
    bn += HEAD_FILTERS
    return {"trainable": w + 2 * bn, "total": w + 4 * bn}


def count_macs(blocks: Optional[List[OBlock]] = None, crop: int = 256, n_head_logits: int = 288) -> int:
    """Multiply-accumulates for one crop: stem + blocks + 1x1 head conv + the MetrABS Linear(1280, 288) pose head."""
    blocks = oracle_blocks(crop // 2) if blocks is None else blocks
    m = (crop // 2) ** 2 * 27 * STEM_FILTERS
    for b in blocks:
        o = b.out_hw * b.out_hw
        if b.kind == "fused":
            m += o * 9 * b.cin * b.cexp + (o * b.cexp * b.cout if b.cexp != b.cin else 0)
        else:
            m += b.in_hw * b.in_hw * b.cin * b.cexp + o * 9 * b.cexp + o * b.cexp * b.cout + 2 * b.cexp * b.cse
    o = blocks[-1].out_hw ** 2
    return m + o * blocks[-1].cout * HEAD_FILTERS + o * HEAD_FILTERS * n_head_logits


def _r(x: torch.Tensor, mode: str) -> torch.Tensor:
    """Storage rounding: "bf16" = one bf16 value; "bf16x2" = hi + lo bf16 pair (x ~ hi + bf16(x - hi), 16 mantissa
    bits: what an f32-grade residual stream stores); "f32" = none."""
    if mode == "bf16":
        return x.bfloat16().float()
    if mode == "bf16x2":
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    if mode == "f16":
        return x.clamp(-65504.0, 65504.0).half().float()      # the device saturates instead of producing inf
    return x


def _silu(x):
    return x * torch.sigmoid(x)


ROUND_POINTS = ("w", "expand", "dw", "gate", "out")


class EffNetV2LOracle:
    """mode "f32" / "bf16" as in the module docstring. ``rounding`` (optional) overrides the storage type per
    rounding point: a callable (stage, point) -> "f32" | "bf16" | "f16" | "bf16x2" (hi + lo pair) with stage = index of the block string
    (0..6; -1 = stem output, 7 = the 640->1280 conv's weights) and point in ROUND_POINTS:
        "w"       conv + depthwise weights of the stage (BN scale folded in)
        "expand"  output of the expand conv (3x3 for Fused-MBConv, 1x1 for MBConv)
        "dw"      output of the depthwise conv (after BN + SiLU)
        "gate"    the SE-gated tensor the projection reads
        "out"     the block output = the residual stream ("f32+bf16copy": the skip path keeps f32, the next block's
                  convolutions read a bf16 copy -- a layout the budget evaluates, not one the product has)
    This is what oracle/error_budget.py switches one point at a time."""

    F16_FROM = 5        # first block string the product runs in fp16 (hpe_api.cpp f16_from)

    def __init__(self, state: Mapping[str, np.ndarray], mode: str = "bf16", rounding=None):
        assert mode in ("f32", "f16", "bf16", "bf16_plain")
        self.mode = mode
        mixed = rounding is None and mode == "bf16"
        if rounding is None:
            if mode == "bf16":
                rounding = lambda stage, point: "f16" if stage >= self.F16_FROM else "bf16"
            elif mode == "bf16_plain":
                rounding = lambda stage, point: "bf16"
            elif mode == "f16":
                rounding = lambda stage, point: "f16"
            else:
                rounding = lambda stage, point: "f32"
        self.rounding = rounding
        self.blocks = oracle_blocks()
        # storage type of every rounding point of every block: w_expand / w_dw / w_project follow "w"
        self.bt: Dict[int, Dict[str, str]] = {}
        for b in self.blocks:
            t = {pt: rounding(b.stage, pt) for pt in ROUND_POINTS}
            t["w_expand"] = t["w_dw"] = t["w_project"] = t["w"]
            if mixed and b.stage == self.F16_FROM and b.stride == 2:
                # the block that enters the fp16 stages reads the bf16 stream: bf16 expand conv (weights and output) and
                # bf16 depthwise taps; from the depthwise output on it is fp16
                t["w_expand"] = t["expand"] = t["w_dw"] = "bf16"
            self.bt[b.idx] = t
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in state.items():
            self.w[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        # fold the BN scale into the conv weights (what the device does at load), then round
        self.cw: Dict[str, torch.Tensor] = {}
        for k in list(self.w):
            if k.endswith(".w") and k.startswith("bbone.") and ".dw." not in k and ".se." not in k:
                p = k[:-2]
                w = self.w[k] * self.w[p + ".scale"].view(-1, 1, 1, 1)      # [cout,kh,kw,cin]
                if p == "bbone.head":
                    w = _r(w, rounding(7, "w"))
                elif p != "bbone.stem":
                    blk, leaf = p.rsplit(".", 1)
                    w = _r(w, self.bt[int(blk.split(".b")[1])]["w_" + leaf])
                self.cw[p] = w.permute(0, 3, 1, 2).contiguous()             # -> OIHW
        for b in self.blocks:
            if b.kind == "mb":
                p = f"bbone.b{b.idx}.dw"
                self.cw[p] = _r(self.w[p + ".w"] * self.w[p + ".scale"].view(-1, 1, 1),
                                self.bt[b.idx]["w_dw"]).unsqueeze(1)        # [c,1,3,3], rounded taps

    def _conv(self, x, p, k, stride, act, out_f32=False):
        w = self.cw[p]
        if k == 3:
            if stride == 2:
                x = F.pad(x, (0, 1, 0, 1))                                  # TF SAME, even input
                y = F.conv2d(x, w, stride=2)
            else:
                y = F.conv2d(x, w, padding=1)
        else:
            y = F.conv2d(x, w)
        y = y + self.w[p + ".shift"].view(1, -1, 1, 1)
        if act:
            y = _silu(y)
        return y if out_f32 else y

    def backbone(self, crops_nhwc: np.ndarray, taps: Optional[dict] = None) -> np.ndarray:
        """crops [B,256,256,3] f32 in [0,1] -> features [B,8,8,1280] f32."""
        rd = self.rounding
        x = torch.from_numpy(np.ascontiguousarray(crops_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)
        with torch.no_grad():
            x = _r(self._conv(x, "bbone.stem", 3, 2, True), rd(-1, "out"))  # stem runs in f32 on f32 crops
            if taps is not None:
                taps["stem"] = x.permute(0, 2, 3, 1).numpy().copy()
            skip = x          # the residual stream; equals x unless it is kept in f32 beside a bf16 copy
            for b in self.blocks:
                p = f"bbone.b{b.idx}"
                bt = self.bt[b.idx]
                if b.kind == "fused":
                    if b.cexp == b.cin:
                        y = self._conv(x, p + ".expand", 3, b.stride, True)
                    else:
                        h = _r(self._conv(x, p + ".expand", 3, b.stride, True), bt["expand"])
                        y = self._conv(h, p + ".project", 1, 1, False)
                else:
                    h = _r(self._conv(x, p + ".expand", 1, 1, True), bt["expand"])
                    wd = self.cw[p + ".dw"]
                    if b.stride == 2:
                        d = F.conv2d(F.pad(h, (0, 1, 0, 1)), wd, stride=2, groups=b.cexp)
                    else:
                        d = F.conv2d(h, wd, padding=1, groups=b.cexp)
                    d = _r(_silu(d + self.w[p + ".dw.shift"].view(1, -1, 1, 1)), bt["dw"])
                    s = d.mean(dim=(2, 3))                                                    # [B,c] f32
                    s = _silu(s @ self.w[p + ".se.w1"].T + self.w[p + ".se.b1"])
                    g = torch.sigmoid(s @ self.w[p + ".se.w2"].T + self.w[p + ".se.b2"])      # [B,c]
                    h2 = _r(d * g.view(g.shape[0], -1, 1, 1), bt["gate"])
                    y = self._conv(h2, p + ".project", 1, 1, False)
                if b.residual:
                    y = y + skip
                if bt["out"] == "f32+bf16copy":
                    skip = y                                   # f32 residual stream ...
                    x = _r(y, "bf16")                          # ... and the bf16 copy the next block's convs read
                else:
                    x = skip = _r(y, bt["out"])
                if taps is not None and (b.idx in taps.get("_want", ()) or taps.get("_all")):
                    taps[f"b{b.idx}"] = x.permute(0, 2, 3, 1).numpy().copy()
            x = self._conv(x, "bbone.head", 1, 1, True)                     # stored f32
        return x.permute(0, 2, 3, 1).contiguous().numpy()

    def head(self, feat: np.ndarray) -> np.ndarray:
        """features [B,8,8,1280] -> pose-head logits [B,8,8,288] (4_create_heads_onnx.py:10-15)"""
        return feat @ self.w["head.weight"].numpy().T + self.w["head.bias"].numpy()
