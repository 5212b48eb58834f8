"""ORACLE (test infrastructure, NOT product code) -- PARITY UNPINNED.

CPU definition (PyTorch fp32 ops) of the ResNet-50 trunk the reference builds for its RGB / hybrid input types as
``nn.Sequential(*list(torchvision.models.resnet50(pretrained=True).children())[:-1])`` (modules/ar/utils/model.py:270-277):
images [N,3,224,224] -> [N,2048] ("trunk features", the input of PostResNet, model.py:207-216). torchvision and its
pretrained weights are not in the reference tree, so there is no reference output to pin against: this file restates the
PUBLIC torchvision architecture from its layer configuration (Bottleneck, [3, 4, 6, 3], expansion 4, stride on the 3x3
convolution, BatchNorm eps 1e-5) -- nothing is imported from the product package -- and the HIP trunk is compared with it on
synthetic weights. Known answers asserted on this file's own table in tests/test_rgb_cpu.py: 23,508,032 trainable
parameters without the classifier (torchvision's resnet50 has 25,557,032 with its 2,049,000-parameter fc) and 4.09 GMAC per
224 x 224 image (with the fc's 2.05 M).

Modes: "f32" plain fp32; "bf16" the storage points of the HIP path (conv weights with the BN scale folded in and every stored
activation rounded to bf16 -- the 7x7 stem's weights stay f32 like the image --, accumulation / bias / ReLU / pools in fp32).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LAYER_CFG = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))     # (planes, blocks, stride) of layer1..4
EXPANSION = 4


def oracle_blocks(img: int = 224) -> List[dict]:
    out, hw, cin = [], img // 4, 64
    for li, (planes, n, stride) in enumerate(LAYER_CFG, start=1):
        for i in range(n):
            st = stride if i == 0 else 1
            out.append(dict(name=f"rgb.layer{li}.{i}", cin=cin, planes=planes, stride=st, in_hw=hw, out_hw=hw // st,
                            down=(i == 0 and (st != 1 or cin != planes * EXPANSION))))
            hw //= st
            cin = planes * EXPANSION
    return out


def count_parameters() -> int:
    n = 64 * 3 * 49 + 2 * 64
    for b in oracle_blocks():
        p, c = b["planes"], b["cin"]
        n += p * c + 2 * p + 9 * p * p + 2 * p + 4 * p * p + 2 * 4 * p
        if b["down"]:
            n += 4 * p * c + 2 * 4 * p
    return n


def count_macs(img: int = 224) -> int:
    m = (img // 2) ** 2 * 64 * 147
    for b in oracle_blocks(img):
        p, c = b["planes"], b["cin"]
        m += b["in_hw"] ** 2 * c * p + b["out_hw"] ** 2 * (9 * p * p + 4 * p * p)
        if b["down"]:
            m += b["out_hw"] ** 2 * c * 4 * p
    return m


def _r(x, mode):
    return x.bfloat16().float() if mode == "bf16" else x


class ResNet50Oracle:
    def __init__(self, state: Mapping[str, np.ndarray], mode: str = "bf16"):
        assert mode in ("f32", "bf16")
        self.mode = mode
        self.blocks = oracle_blocks()
        self.w: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        for k, v in state.items():
            if k.endswith(".w"):
                p = k[:-2]
                w = torch.from_numpy(np.ascontiguousarray(v, np.float32)) * torch.from_numpy(np.asarray(state[p + ".scale"], np.float32)).view(-1, 1, 1, 1)
                if p != "rgb.conv1":
                    w = _r(w, mode)
                self.w[p] = (w.permute(0, 3, 1, 2).contiguous(), torch.from_numpy(np.asarray(state[p + ".shift"], np.float32)))

    def _conv(self, x, p, stride=1, pad=0):
        w, b = self.w[p]
        return F.conv2d(x, w, stride=stride, padding=pad) + b.view(1, -1, 1, 1)

    def forward(self, images_nchw: np.ndarray) -> np.ndarray:
        m = self.mode
        x = torch.from_numpy(np.ascontiguousarray(images_nchw, np.float32))
        with torch.no_grad():
            x = _r(F.relu(self._conv(x, "rgb.conv1", 2, 3)), m)
            x = F.max_pool2d(x, 3, 2, 1)
            for b in self.blocks:
                p = b["name"]
                h = _r(F.relu(self._conv(x, p + ".conv1")), m)
                h = _r(F.relu(self._conv(h, p + ".conv2", b["stride"], 1)), m)
                skip = _r(self._conv(x, p + ".downsample", b["stride"]), m) if b["down"] else x
                x = _r(F.relu(self._conv(h, p + ".conv3") + skip), m)
            x = x.mean(dim=(2, 3))
        return x.numpy()
