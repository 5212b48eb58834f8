"""Thin object wrappers over the C ABI handles (include/isbfsar.h).

numpy arrays go through the ``*_host`` entry points (H2D + kernels + D2H, like the reference's
``Runner.__call__``, utils/tensorrt_runner.py:64-77); torch CUDA tensors are passed by device
pointer on torch's current stream -- torch is only the owner of device memory and streams here.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional, Tuple, Union

import numpy as np

from . import _lib
from .weights import pack_blob


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32c(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(out.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(out.shape)}")
    return out


class ArEngine:
    """TRXOS (skeleton branch + Discriminator) on one MI355X. Mirrors ``TRXOS.forward``
    (reference modules/ar/utils/model.py:291-328) for B windows sharing one support set."""

    def __init__(self, seq_len: int, n_joints: int, way_max: int, device: int = 0,
                 precision: Union[int, str] = "default", max_batch: int = 1024, input_type: str = "skeleton"):
        """input_type (TRXConfig.input_type): "skeleton" (features = MLP(pose), 256 wide) or "hybrid" (features =
        [PostResNet(ResNet-50 trunk) | MLP(pose)], 512 wide; set_support / infer then also take trunk features
        [.., L, 2048], e.g. from RgbEngine)."""
        # "default" = ISB_AR_PREC_DEFAULT = what a zero-initialised isb_ar_cfg gets: fp16 operands (include/isbfsar.h); the
        # resolved setting is read back from the handle (isb_ar_precision), so that there is ONE default, the library's
        names = {"default": _lib.ISB_AR_PREC_DEFAULT, "bf16": _lib.ISB_AR_PREC_BF16, "bf16x3": _lib.ISB_AR_PREC_BF16X3,
                 "f16": _lib.ISB_AR_PREC_F16}
        if isinstance(precision, str):
            if precision not in names:
                raise ValueError(f"precision {precision!r} not in {sorted(names)}")
            prec = names[precision]
        else:
            # a raw integer is an ABI constant. 0 changed its meaning with ABI version 2 (it was bf16, it is "the library's default" =
            # fp16 operands now; bf16 moved to 3): a caller written against version 1 must notice, so 0 is refused here -- say
            # "default" or "bf16" (ADVICE r5). The C ABI keeps accepting 0 (a zero-initialised isb_ar_cfg).
            if int(precision) == 0:
                raise ValueError('precision=0 is ambiguous across ABI versions (v1: bf16, v2: the default = fp16 operands): '
                                 'pass "default", "f16", "bf16" or "bf16x3"')
            if int(precision) not in (1, 2, 3):
                raise ValueError(f"precision {precision!r}: ISB_AR_PREC_BF16X3 (1), ISB_AR_PREC_F16 (2) or ISB_AR_PREC_BF16 (3)")
            prec = int(precision)
        if input_type not in ("skeleton", "hybrid"):
            raise ValueError(f"input_type {input_type!r}: 'skeleton' or 'hybrid' (the reference's 'rgb' type is inconsistent with "
                             "its own model: utils/params.py:81 sizes the transformer for 1000-wide features, model.py:274-277 makes 256)")
        self.L, self.J, self.way_max, self.device = seq_len, n_joints, way_max, device
        self.input_type = input_type
        self.d_in = 512 if input_type == "hybrid" else 256
        self.n = 0
        self._h = C.c_void_p()
        cfg = _lib.isb_ar_cfg(seq_len, n_joints, way_max, device, prec, max_batch)
        _lib.check(_lib.lib().isb_ar_create(C.byref(cfg), C.byref(self._h)), "isb_ar_create")
        self.precision = {_lib.ISB_AR_PREC_BF16X3: "bf16x3", _lib.ISB_AR_PREC_F16: "f16",
                          _lib.ISB_AR_PREC_BF16: "bf16"}[_lib.lib().isb_ar_precision(self._h)]
        if input_type == "hybrid":
            _lib.check(_lib.lib().isb_ar_set_input_type(self._h, 1), "isb_ar_set_input_type")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().isb_ar_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------
    def load_weights(self, state: Union[bytes, Mapping[str, np.ndarray]]):
        blob = state if isinstance(state, (bytes, bytearray)) else pack_blob(state)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _lib.check(_lib.lib().isb_ar_load_weights(self._h, C.cast(buf, C.c_void_p), len(blob)),
                   "isb_ar_load_weights")
        self.n = 0

    # -- support set ---------------------------------------------------------------------
    def set_support(self, poses: Optional[np.ndarray] = None, features: Optional[np.ndarray] = None,
                    trunk: Optional[np.ndarray] = None):
        """poses [n,L,3J] (hybrid: with trunk [n,L,2048]) or cached features [n,L,d_in]"""
        if (poses is None) == (features is None):
            raise ValueError(f"give exactly one of poses [n,L,3J] / features [n,L,{self.d_in}]")
        if poses is not None:
            poses = _f32c(poses)
            n = poses.shape[0]
            _f32c(poses, (n, self.L, 3 * self.J))
            if self.input_type == "hybrid":
                if trunk is None:
                    raise ValueError("hybrid input type: poses need the RGB trunk features [n,L,2048] beside them")
                trunk = _f32c(trunk, (n, self.L, 2048))
                _lib.check(_lib.lib().isb_ar_set_support_hybrid(self._h, _ptr(poses), _ptr(trunk), n), "isb_ar_set_support_hybrid")
                self.n = n
                return
        else:
            features = _f32c(features)
            n = features.shape[0]
            _f32c(features, (n, self.L, self.d_in))
        _lib.check(_lib.lib().isb_ar_set_support(self._h, _ptr(poses), _ptr(features), n), "isb_ar_set_support")
        self.n = n

    def support_features(self) -> np.ndarray:
        out = np.empty((self.n, self.L, self.d_in), np.float32)
        _lib.check(_lib.lib().isb_ar_get_support_features(self._h, _ptr(out)), "isb_ar_get_support_features")
        return out

    # -- inference -----------------------------------------------------------------------
    def infer(self, windows, want_embed: bool = False, trunk=None):
        """windows [B,L,3J]: numpy -> numpy results; torch CUDA tensor -> torch CUDA results
        (asynchronous on the current stream). Returns (logits [B,n], is_true [B], embed|None).
        hybrid input type: trunk [B,L,2048] (same kind of array as windows) beside the windows; embed is 512 wide."""
        if self.input_type == "hybrid":
            return self._infer_hybrid(windows, trunk, want_embed)
        if isinstance(windows, np.ndarray):
            w = _f32c(windows)
            B = w.shape[0]
            _f32c(w, (B, self.L, 3 * self.J))
            logits = np.empty((B, self.n), np.float32)
            is_true = np.empty((B,), np.float32)
            embed = np.empty((B, self.L, 256), np.float32) if want_embed else None
            _lib.check(_lib.lib().isb_ar_infer_host(self._h, _ptr(w), B, _ptr(logits), _ptr(is_true), _ptr(embed)),
                       "isb_ar_infer_host")
            return logits, is_true, embed
        import torch

        if not (isinstance(windows, torch.Tensor) and windows.is_cuda):
            raise TypeError("windows must be a numpy array or a torch CUDA tensor")
        if windows.device.index != self.device:
            raise ValueError(f"windows live on cuda:{windows.device.index}, engine on cuda:{self.device}")
        w = windows.contiguous().float()
        B = w.shape[0]
        if tuple(w.shape) != (B, self.L, 3 * self.J):
            raise ValueError(f"expected [B,{self.L},{3 * self.J}], got {tuple(w.shape)}")
        logits = torch.empty((B, self.n), dtype=torch.float32, device=w.device)
        is_true = torch.empty((B,), dtype=torch.float32, device=w.device)
        embed = torch.empty((B, self.L, 256), dtype=torch.float32, device=w.device) if want_embed else None
        stream = torch.cuda.current_stream(w.device).cuda_stream
        _lib.check(_lib.lib().isb_ar_infer(self._h, w.data_ptr(), B, logits.data_ptr(), is_true.data_ptr(),
                                           embed.data_ptr() if want_embed else None, C.c_void_p(stream)),
                   "isb_ar_infer")
        return logits, is_true, embed

    def _infer_hybrid(self, windows, trunk, want_embed):
        import torch
        if trunk is None:
            raise ValueError("hybrid input type: infer() needs trunk [B,L,2048]")
        host = isinstance(windows, np.ndarray)
        dev = torch.device("cuda", self.device)
        w = (torch.from_numpy(_f32c(windows)).to(dev) if host else windows).contiguous().float()
        t = (torch.from_numpy(_f32c(trunk)).to(dev) if isinstance(trunk, np.ndarray) else trunk).contiguous().float()
        B = w.shape[0]
        if tuple(w.shape) != (B, self.L, 3 * self.J) or tuple(t.shape) != (B, self.L, 2048):
            raise ValueError(f"expected [B,{self.L},{3 * self.J}] and [B,{self.L},2048], got {tuple(w.shape)} {tuple(t.shape)}")
        logits = torch.empty((B, self.n), dtype=torch.float32, device=dev)
        is_true = torch.empty((B,), dtype=torch.float32, device=dev)
        embed = torch.empty((B, self.L, 512), dtype=torch.float32, device=dev) if want_embed else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().isb_ar_infer_hybrid(self._h, w.data_ptr(), t.data_ptr(), B, logits.data_ptr(), is_true.data_ptr(),
                                                  embed.data_ptr() if want_embed else None, C.c_void_p(stream)), "isb_ar_infer_hybrid")
        if host:
            torch.cuda.synchronize(dev)
            return logits.cpu().numpy(), is_true.cpu().numpy(), None if embed is None else embed.cpu().numpy()
        return logits, is_true, embed

    def last_chosen(self, B: int) -> np.ndarray:
        out = np.empty((B,), np.int32)
        _lib.check(_lib.lib().isb_ar_last_chosen(self._h, _ptr(out), B), "isb_ar_last_chosen")
        return out

    # -- profiling of the tuple-attention kernels (bench.py roofline) ------------------------
    def profile(self, enable: bool):
        _lib.check(_lib.lib().isb_ar_profile(self._h, int(enable)), "isb_ar_profile")

    def profile_read(self) -> Tuple[float, int]:
        ms = C.c_double()
        n = C.c_int64()
        _lib.check(_lib.lib().isb_ar_profile_read(self._h, C.byref(ms), C.byref(n)), "isb_ar_profile_read")
        return ms.value, n.value
