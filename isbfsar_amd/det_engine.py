"""Object wrapper over the isb_det_* entry points (include/isbfsar.h): the YOLOv4 person detector the reference runs as
``Runner(model_config.yolo_engine_path)`` (modules/hpe/hpe.py:42,51-60)."""
from __future__ import annotations

import ctypes as C
from typing import List, Mapping, Optional, Tuple, Union

import numpy as np

from . import _lib
from .weights import pack_blob

N_BOXES, N_CLASSES = 4032, 80


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def describe_convs() -> List[Tuple[str, int, int, int, int, int, bool]]:
    """(name, cin, cout, k, stride, act, has_batchnorm) of every convolution, from the library's own plan."""
    lib = _lib.lib()
    out = []
    buf = C.create_string_buffer(128)
    dims = (C.c_int32 * 6)()
    for i in range(lib.isb_det_n_convs()):
        _lib.check(lib.isb_det_describe(i, buf, 128, dims), "isb_det_describe")
        out.append((buf.value.decode(), dims[0], dims[1], dims[2], dims[3], dims[4], bool(dims[5])))
    return out


class DetEngine:
    """frames -> (boxes [B,4032,1,4], confs [B,4032,80]), what hpe.py:60 reshapes the YOLO engine's outputs to."""

    def __init__(self, width: int = 640, height: int = 480, device: int = 0, max_batch: int = 16):
        self.width, self.height, self.device, self.max_batch = width, height, device, max_batch
        self._h = C.c_void_p()
        cfg = _lib.isb_det_cfg(width, height, device, max_batch)
        _lib.check(_lib.lib().isb_det_create(C.byref(cfg), C.byref(self._h)), "isb_det_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().isb_det_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, state: Union[bytes, Mapping[str, np.ndarray]]):
        blob = state if isinstance(state, (bytes, bytearray)) else pack_blob(state)
        buf = np.frombuffer(blob, dtype=np.uint8)
        _lib.check(_lib.lib().isb_det_load_weights(self._h, _ptr(buf), len(blob)), "isb_det_load_weights")

    def forward(self, frames):
        """frames uint8 [B,H,W,3] BGR. numpy -> numpy; torch CUDA -> torch CUDA (asynchronous on the current stream)."""
        if isinstance(frames, np.ndarray):
            f = np.ascontiguousarray(frames, dtype=np.uint8)
            B = f.shape[0]
            if f.shape != (B, self.height, self.width, 3):
                raise ValueError(f"bad frame shape {f.shape}")
            boxes = np.empty((B, N_BOXES, 1, 4), np.float32)
            confs = np.empty((B, N_BOXES, N_CLASSES), np.float32)
            _lib.check(_lib.lib().isb_det_forward_host(self._h, _ptr(f), B, _ptr(boxes), _ptr(confs)), "isb_det_forward_host")
            return boxes, confs
        import torch
        if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8):
            raise TypeError("frames must be a uint8 numpy array or a uint8 torch CUDA tensor")
        f = frames.contiguous()
        B = f.shape[0]
        if tuple(f.shape) != (B, self.height, self.width, 3):
            raise ValueError(f"bad frame shape {tuple(f.shape)}")
        boxes = torch.empty((B, N_BOXES, 1, 4), dtype=torch.float32, device=f.device)
        confs = torch.empty((B, N_BOXES, N_CLASSES), dtype=torch.float32, device=f.device)
        stream = torch.cuda.current_stream(f.device).cuda_stream
        _lib.check(_lib.lib().isb_det_forward(self._h, f.data_ptr(), B, boxes.data_ptr(), confs.data_ptr(), C.c_void_p(stream)),
                   "isb_det_forward")
        return boxes, confs

    def debug(self, frames: np.ndarray):
        """(image f32 [B,256,256,3] RGB, maps: three f32 [B,H,W,255]) -- stage hook for the tests, B <= max_batch."""
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        B = f.shape[0]
        img = np.empty((B, 256, 256, 3), np.float32)
        maps = [np.empty((B, hw, hw, 256), np.float32) for hw in (32, 16, 8)]
        _lib.check(_lib.lib().isb_det_debug_host(self._h, _ptr(f), B, _ptr(img), _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2])),
                   "isb_det_debug_host")
        return img, [m[..., :255] for m in maps]
