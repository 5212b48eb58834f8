"""Object wrapper over the isb_rgb_* entry points (include/isbfsar.h): the ResNet-50 trunk of the hybrid input type."""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Union

import numpy as np

from . import _lib
from .weights import pack_blob


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RgbEngine:
    """images f32 [N,3,224,224] (the layout main.py:91 produces) or [N,224,224,3] -> trunk features f32 [N,2048]
    (modules/ar/utils/model.py:270-277, 297)."""

    def __init__(self, device: int = 0, max_batch: int = 64):
        self.device = device
        self._h = C.c_void_p()
        cfg = _lib.isb_rgb_cfg(device, max_batch)
        _lib.check(_lib.lib().isb_rgb_create(C.byref(cfg), C.byref(self._h)), "isb_rgb_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().isb_rgb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, state: Union[bytes, Mapping[str, np.ndarray]]):
        blob = state if isinstance(state, (bytes, bytearray)) else pack_blob(state)
        buf = np.frombuffer(blob, dtype=np.uint8)
        _lib.check(_lib.lib().isb_rgb_load_weights(self._h, _ptr(buf), len(blob)), "isb_rgb_load_weights")

    def forward(self, images, nchw: bool = True):
        shape = (3, 224, 224) if nchw else (224, 224, 3)
        if isinstance(images, np.ndarray):
            x = np.ascontiguousarray(images, dtype=np.float32)
            if tuple(x.shape[1:]) != shape:
                raise ValueError(f"expected [N,{shape}], got {x.shape}")
            out = np.empty((x.shape[0], 2048), np.float32)
            _lib.check(_lib.lib().isb_rgb_forward_host(self._h, _ptr(x), x.shape[0], int(nchw), _ptr(out)), "isb_rgb_forward_host")
            return out
        import torch
        if not (isinstance(images, torch.Tensor) and images.is_cuda):
            raise TypeError("images must be a numpy array or a torch CUDA tensor")
        x = images.contiguous().float()
        if tuple(x.shape[1:]) != shape:
            raise ValueError(f"expected [N,{shape}], got {tuple(x.shape)}")
        out = torch.empty((x.shape[0], 2048), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().isb_rgb_forward(self._h, x.data_ptr(), x.shape[0], int(nchw), out.data_ptr(), C.c_void_p(stream)),
                   "isb_rgb_forward")
        return out
