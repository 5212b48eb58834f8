"""ORACLE (test infrastructure, NOT product code) -- PARITY UNPINNED.

CPU definition (PyTorch fp32 ops) of the EfficientNetV2-L backbone the reference runs as an opaque
TensorRT engine (``bbone1.engine``: reference utils/params.py:29, modules/hpe/hpe.py:103; contract
``f32[B,256,256,3]`` NHWC in [0,1] -> ``f32[B,8,8,1280]``: modules/hpe/setup/7_create_engines.py:38-42,
4_create_heads_onnx.py:13,19; model name 'efficientnetv2-l', include_top=False:
2_extract_bbone_heads.py:27,46-47). The arithmetic lives in isarandi/metrabs (un-vendored, no pinned
version) + downloaded weights, so there is NO reference output to pin against: this file restates
the public efficientnetv2-l block table (isbfsar_amd/effnetv2.py) and the HIP backbone is compared
with it on synthetic weights. What IS pinned is the shape contract.

Two numeric modes:
  * ``mode="f32"``   plain fp32 everywhere;
  * ``mode="bf16"``  the storage/rounding points of the HIP path: conv weights incl. the depthwise taps (BN
    scale folded in) and every stored activation are rounded to bf16, accumulation / bias / SiLU / SE in fp32, the
    last 1x1 conv (640->1280) stores f32 and the pose head runs in f32 -- so a GPU-vs-oracle
    difference is accumulation order only.
Only tests/, smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

from isbfsar_amd import effnetv2 as arch


def _r(x: torch.Tensor, mode: str) -> torch.Tensor:
    return x.bfloat16().float() if mode == "bf16" else x


def _silu(x):
    return x * torch.sigmoid(x)


class EffNetV2LOracle:
    def __init__(self, state: Mapping[str, np.ndarray], mode: str = "bf16"):
        assert mode in ("f32", "bf16")
        self.mode = mode
        self.blocks = arch.blocks()
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in state.items():
            self.w[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        # fold the BN scale into the conv weights (what the device does at load), then round
        self.cw: Dict[str, torch.Tensor] = {}
        for k in list(self.w):
            if k.endswith(".w") and k.startswith("bbone.") and ".dw." not in k and ".se." not in k:
                p = k[:-2]
                w = self.w[k] * self.w[p + ".scale"].view(-1, 1, 1, 1)      # [cout,kh,kw,cin]
                if p != "bbone.stem":
                    w = _r(w, mode)
                self.cw[p] = w.permute(0, 3, 1, 2).contiguous()             # -> OIHW
        for b in self.blocks:
            if b.kind == "mb":
                p = f"bbone.b{b.idx}.dw"
                self.cw[p] = _r(self.w[p + ".w"] * self.w[p + ".scale"].view(-1, 1, 1), mode).unsqueeze(1)  # [c,1,3,3], bf16-rounded taps

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
        m = self.mode
        x = torch.from_numpy(np.ascontiguousarray(crops_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)
        with torch.no_grad():
            x = _r(self._conv(x, "bbone.stem", 3, 2, True), m)              # stem runs in f32 on f32 crops
            if taps is not None:
                taps["stem"] = x.permute(0, 2, 3, 1).numpy().copy()
            for b in self.blocks:
                p = f"bbone.b{b.idx}"
                inp = x
                if b.kind == "fused":
                    if b.cexp == b.cin:
                        y = self._conv(x, p + ".expand", 3, b.stride, True)
                    else:
                        h = _r(self._conv(x, p + ".expand", 3, b.stride, True), m)
                        y = self._conv(h, p + ".project", 1, 1, False)
                else:
                    h = _r(self._conv(x, p + ".expand", 1, 1, True), m)
                    wd = self.cw[p + ".dw"]
                    if b.stride == 2:
                        d = F.conv2d(F.pad(h, (0, 1, 0, 1)), wd, stride=2, groups=b.cexp)
                    else:
                        d = F.conv2d(h, wd, padding=1, groups=b.cexp)
                    d = _r(_silu(d + self.w[p + ".dw.shift"].view(1, -1, 1, 1)), m)
                    s = d.mean(dim=(2, 3))                                                    # [B,c] f32
                    s = _silu(s @ self.w[p + ".se.w1"].T + self.w[p + ".se.b1"])
                    g = torch.sigmoid(s @ self.w[p + ".se.w2"].T + self.w[p + ".se.b2"])      # [B,c]
                    h2 = _r(d * g.view(g.shape[0], -1, 1, 1), m)
                    y = self._conv(h2, p + ".project", 1, 1, False)
                if b.residual:
                    y = y + inp
                x = _r(y, m)
                if taps is not None and (b.idx in taps.get("_want", ()) or taps.get("_all")):
                    taps[f"b{b.idx}"] = x.permute(0, 2, 3, 1).numpy().copy()
            x = self._conv(x, "bbone.head", 1, 1, True)                     # stored f32
        return x.permute(0, 2, 3, 1).contiguous().numpy()

    def head(self, feat: np.ndarray) -> np.ndarray:
        """features [B,8,8,1280] -> pose-head logits [B,8,8,288] (4_create_heads_onnx.py:10-15)"""
        return feat @ self.w["head.weight"].numpy().T + self.w["head.bias"].numpy()
