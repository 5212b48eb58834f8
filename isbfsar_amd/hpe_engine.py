"""Object wrapper over the isb_hpe_* / isb_pose_windows entry points (include/isbfsar.h)."""
from __future__ import annotations

import ctypes as C
import json
from typing import Mapping, Optional, Sequence, Union

import numpy as np

from . import _lib
from .weights import pack_blob



PRECISIONS = {"f16": 2, "bf16": 1, "bf16_f16tail": 3}      # isb_hpe_cfg.precision (0 = the library's default = "f16")


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HpeEngine:
    """Everything ``HumanPoseEstimator.estimate`` does after the detector (reference
    modules/hpe/hpe.py:76-173), batched, on one MI355X."""

    def __init__(self, fx=384.025146484375, fy=384.025146484375, ppx=319.09661865234375,
                 ppy=237.75723266601562, width=640, height=480, device: int = 0, max_batch: int = 64,
                 precision: str = "f16"):
        """precision (16-bit storage of the backbone's weights and activations, f32 accumulate): "f16" (default: IEEE fp16 in
        every stage -- what the reference's TensorRT engines run, 7_create_engines.py:10, and the closest 16-bit layout to the
        fp32 definition), "bf16" (bf16 everywhere) or "bf16_f16tail" (round 3's layout: bf16 with the two
        8x8 stages and the 640 -> 1280 convolution in fp16)."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision {precision!r} not in {sorted(PRECISIONS)}")
        self.width, self.height, self.device, self.max_batch = width, height, device, max_batch
        self.precision = precision
        self.n_out = 0
        self._h = C.c_void_p()
        cfg = _lib.isb_hpe_cfg(fx, fy, ppx, ppy, width, height, device, max_batch, 0, PRECISIONS[precision])
        _lib.check(_lib.lib().isb_hpe_create(C.byref(cfg), C.byref(self._h)), "isb_hpe_create")

    def share(self) -> "HpeEngine":
        """One more engine on THIS engine's device weights (isb_hpe_create_shared): the child reads this engine's model (weights, joint
        map) and owns its own streams and workspaces. For callers that keep several batches in flight, one engine per batch in flight,
        each forward() on its own stream: one copy of the 240 MB model + K workspaces instead of K of each, same bits."""
        child = HpeEngine.__new__(HpeEngine)
        child.width, child.height, child.device, child.max_batch = self.width, self.height, self.device, self.max_batch
        child.precision, child.n_out = self.precision, self.n_out
        child._h = C.c_void_p()
        child._parent = self          # (keeps the Python object of the parent alive; the library itself does not need it)
        _lib.check(_lib.lib().isb_hpe_create_shared(self._h, C.byref(child._h)), "isb_hpe_create_shared")
        return child

    def memory(self):
        """(bytes of the model this engine reads, bytes of the workspaces it owns, engines that share the model)"""
        m, w, n = C.c_uint64(), C.c_uint64(), C.c_int32()
        _lib.check(_lib.lib().isb_hpe_memory(self._h, C.byref(m), C.byref(w), C.byref(n)), "isb_hpe_memory")
        return int(m.value), int(w.value), int(n.value)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().isb_hpe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, state: Union[bytes, Mapping[str, np.ndarray]]):
        blob = state if isinstance(state, (bytes, bytearray)) else pack_blob(state)
        buf = np.frombuffer(blob, dtype=np.uint8)
        _lib.check(_lib.lib().isb_hpe_load_weights(self._h, _ptr(buf), len(blob)), "isb_hpe_load_weights")

    def set_joint_map(self, expand: np.ndarray, indices: Optional[Sequence[int]]):
        e = np.ascontiguousarray(expand, dtype=np.float32)
        if e.shape != (32, 122):
            raise ValueError(f"expand must be [32,122], got {e.shape}")
        idx = None if indices is None else np.ascontiguousarray(indices, dtype=np.int32)
        n = 122 if idx is None else int(idx.shape[0])
        _lib.check(_lib.lib().isb_hpe_set_joint_map(self._h, _ptr(e), _ptr(idx), n), "isb_hpe_set_joint_map")
        self.n_out = n

    # -- full stage -----------------------------------------------------------------------
    def forward(self, frames, bboxes):
        """frames uint8 [B,H,W,3] BGR, bboxes int32 [B,4] (x1,x2,y1,y2). numpy -> numpy, torch CUDA
        -> torch CUDA (asynchronous). Returns (joints f32 [B,n_out,3], valid u8 [B])."""
        if isinstance(frames, np.ndarray):
            f = np.ascontiguousarray(frames, dtype=np.uint8)
            bb = np.ascontiguousarray(bboxes, dtype=np.int32)
            B = f.shape[0]
            if f.shape != (B, self.height, self.width, 3) or bb.shape != (B, 4):
                raise ValueError(f"bad shapes {f.shape} {bb.shape}")
            joints = np.empty((B, self.n_out, 3), np.float32)
            valid = np.empty((B,), np.uint8)
            _lib.check(_lib.lib().isb_hpe_forward_host(self._h, _ptr(f), _ptr(bb), B, _ptr(joints), _ptr(valid)),
                       "isb_hpe_forward_host")
            return joints, valid
        import torch
        if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8):
            raise TypeError("frames must be a uint8 numpy array or a uint8 torch CUDA tensor")
        f = frames.contiguous()
        bb = bboxes.contiguous()
        B = f.shape[0]
        if tuple(f.shape) != (B, self.height, self.width, 3) or tuple(bb.shape) != (B, 4) or bb.dtype != torch.int32:
            raise ValueError(f"bad shapes/dtypes {tuple(f.shape)} {tuple(bb.shape)} {bb.dtype}")
        joints = torch.empty((B, self.n_out, 3), dtype=torch.float32, device=f.device)
        valid = torch.empty((B,), dtype=torch.uint8, device=f.device)
        stream = torch.cuda.current_stream(f.device).cuda_stream
        _lib.check(_lib.lib().isb_hpe_forward(self._h, f.data_ptr(), bb.data_ptr(), B, joints.data_ptr(),
                                              valid.data_ptr(), C.c_void_p(stream)), "isb_hpe_forward")
        return joints, valid

    def submit(self, frames: np.ndarray, bboxes: np.ndarray):
        """forward() split in two (isb_hpe_submit_host / isb_hpe_wait_host): enqueue a host batch and return at once; `wait()`
        hands back the OLDEST outstanding batch's (joints, valid). Up to two batches in flight: batch k + 1's frames cross PCIe
        while batch k computes. `frames` should be pinned (torch.Tensor.pin_memory().numpy()) and stay untouched until its wait."""
        if not (isinstance(frames, np.ndarray) and frames.dtype == np.uint8 and frames.flags["C_CONTIGUOUS"]):
            raise TypeError("submit takes a contiguous uint8 numpy array")
        bb = np.ascontiguousarray(bboxes, dtype=np.int32)
        B = frames.shape[0]
        if frames.shape != (B, self.height, self.width, 3) or bb.shape != (B, 4):
            raise ValueError(f"bad shapes {frames.shape} {bb.shape}")
        if not hasattr(self, "_inflight"):
            self._inflight = []
        if len(self._inflight) >= 2:
            raise RuntimeError("two batches are in flight: wait() for the oldest first")
        joints = np.empty((B, self.n_out, 3), np.float32)
        valid = np.empty((B,), np.uint8)
        _lib.check(_lib.lib().isb_hpe_submit_host(self._h, _ptr(frames), _ptr(bb), B, _ptr(joints), _ptr(valid)), "isb_hpe_submit_host")
        self._inflight.append((frames, joints, valid))           # keeps the arrays alive until wait()

    def wait(self):
        """(joints f32 [B,n_out,3], valid u8 [B]) of the oldest submitted batch"""
        if not getattr(self, "_inflight", None):
            raise RuntimeError("wait() without a submitted batch")
        _lib.check(_lib.lib().isb_hpe_wait_host(self._h), "isb_hpe_wait_host")
        _, joints, valid = self._inflight.pop(0)
        return joints, valid

    def set_lanes(self, n_lanes: int):
        """1 = whole-batch launches on the caller's stream (for callers that keep several batches in flight on several engines),
        2 = the default split of a batch into two halves on two streams (isb_hpe_set_lanes). Changes no result bit."""
        _lib.check(_lib.lib().isb_hpe_set_lanes(self._h, int(n_lanes)), "isb_hpe_set_lanes")

    # -- stage-level hooks ----------------------------------------------------------------
    def set_augmentations(self, num_aug: int):
        """Test-time augmentation (MetrabsTRTConfig.num_aug, hpe.py:88-93): crop_params / warp then return num_aug
        items per box, [B * num_aug, ...] in (box, augmentation) order. 0 switches it off."""
        self.num_aug = int(num_aug)
        if self.num_aug <= 0:
            self.num_aug = 0
            _lib.check(_lib.lib().isb_hpe_set_augmentations(self._h, 0, None, None), "isb_hpe_set_augmentations")
            return None
        flip, rotflip, gammas, scales = get_augmentations(self.num_aug)
        rf = np.ascontiguousarray(rotflip, dtype=np.float64)
        sc = np.ascontiguousarray(scales, dtype=np.float64)
        _lib.check(_lib.lib().isb_hpe_set_augmentations(self._h, self.num_aug, _ptr(rf), _ptr(sc)), "isb_hpe_set_augmentations")
        return flip, rotflip, gammas, scales

    def crop_params(self, bboxes):
        bb = np.ascontiguousarray(bboxes, dtype=np.int32)
        B = bb.shape[0] * max(getattr(self, "num_aug", 0), 1)
        H = np.empty((B, 3, 3), np.float32)
        newK = np.empty((B, 3, 3), np.float64)
        R = np.empty((B, 3, 3), np.float64)
        _lib.check(_lib.lib().isb_hpe_crop_params_host(self._h, _ptr(bb), bb.shape[0], _ptr(H), _ptr(newK), _ptr(R)),
                   "isb_hpe_crop_params_host")
        return H, newK, R

    def warp(self, frames, bboxes):
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        bb = np.ascontiguousarray(bboxes, dtype=np.int32)
        B = f.shape[0]
        crops = np.empty((B * max(getattr(self, "num_aug", 0), 1), 256, 256, 3), np.float32)
        _lib.check(_lib.lib().isb_hpe_warp_host(self._h, _ptr(f), _ptr(bb), B, _ptr(crops)), "isb_hpe_warp_host")
        return crops

    def backbone(self, crops, want_features=True):
        c = np.ascontiguousarray(crops, dtype=np.float32)
        B = c.shape[0]
        if c.shape != (B, 256, 256, 3):
            raise ValueError(f"crops must be [B,256,256,3], got {c.shape}")
        feat = np.empty((B, 8, 8, 1280), np.float32) if want_features else None
        logits = np.empty((B, 8, 8, 288), np.float32)
        _lib.check(_lib.lib().isb_hpe_backbone_host(self._h, _ptr(c), B, _ptr(feat), _ptr(logits)),
                   "isb_hpe_backbone_host")
        return feat, logits

    def post(self, logits, bboxes, want_pred=False):
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        bb = np.ascontiguousarray(bboxes, dtype=np.int32)
        B = lg.shape[0]
        if lg.shape != (B, 8, 8, 288):
            raise ValueError(f"logits must be [B,8,8,288], got {lg.shape}")
        joints = np.empty((B, self.n_out, 3), np.float32)
        valid = np.empty((B,), np.uint8)
        pred = np.empty((B, 32, 5), np.float64) if want_pred else None
        _lib.check(_lib.lib().isb_hpe_post_host(self._h, _ptr(lg), _ptr(bb), B, _ptr(joints), _ptr(valid), _ptr(pred)),
                   "isb_hpe_post_host")
        return joints, valid, pred

    def select_person(self, boxes, confs, conf_thresh: float = 0.3):
        """YOLOv4 export tensors -> (bbox int32 [B,4] as x1,x2,y1,y2 or -1s, found u8 [B]). numpy -> numpy; torch CUDA
        tensors (DetEngine.forward's) -> torch CUDA tensors, asynchronous on the current stream."""
        if not isinstance(boxes, np.ndarray):
            import torch
            bx, cf = boxes.contiguous().float(), confs.contiguous().float()
            B = bx.shape[0]
            bbox = torch.empty((B, 4), dtype=torch.int32, device=bx.device)
            found = torch.empty((B,), dtype=torch.uint8, device=bx.device)
            stream = torch.cuda.current_stream(bx.device).cuda_stream
            _lib.check(_lib.lib().isb_hpe_select_person(self._h, bx.data_ptr(), cf.data_ptr(), B, conf_thresh, bbox.data_ptr(),
                                                        found.data_ptr(), C.c_void_p(stream)), "isb_hpe_select_person")
            return bbox, found
        bx = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4032, 4)
        cf = np.ascontiguousarray(confs, dtype=np.float32).reshape(-1, 4032, 80)
        B = bx.shape[0]
        bbox = np.empty((B, 4), np.int32)
        found = np.empty((B,), np.uint8)
        _lib.check(_lib.lib().isb_hpe_select_person_host(self._h, _ptr(bx), _ptr(cf), B, conf_thresh, _ptr(bbox), _ptr(found)),
                   "isb_hpe_select_person_host")
        return bbox, found

    def profile(self, enable: bool):
        _lib.check(_lib.lib().isb_hpe_profile(self._h, int(enable)), "isb_hpe_profile")

    def profile_read(self):
        ms = C.c_double()
        n = C.c_int64()
        _lib.check(_lib.lib().isb_hpe_profile_read(self._h, C.byref(ms), C.byref(n)), "isb_hpe_profile_read")
        return ms.value, n.value

    def profile_read_dw(self):
        """(ms, launches) of the stand-alone depthwise launches of the profiled passes"""
        ms = C.c_double()
        n = C.c_int64()
        _lib.check(_lib.lib().isb_hpe_profile_read_dw(self._h, C.byref(ms), C.byref(n)), "isb_hpe_profile_read_dw")
        return ms.value, n.value


def f32_to_f16(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even f32 -> IEEE fp16 bit patterns (uint16)"""
    return np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


def f16_to_f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint16).view(np.float16).astype(np.float32)


def f32_to_bf16(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even f32 -> bf16 bit patterns (uint16)"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(a: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(a, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def conv_debug(x_bf16, w, scale, shift, k, stride, act, res_bf16=None, gate=None, variant=0, iters=1, device=0, f16=False,
               torch_pad=False):
    """One backbone convolution through isb_debug_conv. x_bf16 uint16 [B,H,W,Cin] (bf16 bits),
    w f32 [Cout,k,k,Cin]. Returns (out uint16 [B,OH,OW,Cout], ms_per_launch). f16: x / res / out hold fp16 bits and the
    weights are rounded to fp16 (ConvArgs.f16, the 8x8 stages). torch_pad: a stride-2 3x3 pads symmetrically (PyTorch, the
    detector / ResNet trunk) instead of TF-SAME."""
    x = np.ascontiguousarray(x_bf16, dtype=np.uint16)
    B, H, W, Cin = x.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    Cout = w.shape[0]
    out = np.empty((B, H // stride, W // stride, Cout), np.uint16)
    if f16:
        act = int(act) | 0x100
    ms = C.c_float()
    r = None if res_bf16 is None else np.ascontiguousarray(res_bf16, dtype=np.uint16)
    g = None if gate is None else np.ascontiguousarray(gate, dtype=np.float32)
    _lib.check(_lib.lib().isb_debug_conv(device, _ptr(x), _ptr(w), _ptr(np.ascontiguousarray(scale, dtype=np.float32)),
                                         _ptr(np.ascontiguousarray(shift, dtype=np.float32)), _ptr(r), _ptr(g),
                                         B, H, W, Cin, Cout, k, stride | (0x100 if torch_pad else 0), int(act), variant, iters,
                                         _ptr(out), C.byref(ms)),
               "isb_debug_conv")
    return out, ms.value


def fused_mb_debug(x_bf16, w1, scale1, shift1, w2, scale2, shift2, res_bf16=None, stride=1, iters=1, device=0, f16=False, stamps=False, lds_e=False):
    """A whole Fused-MBConv block through isb_debug_fused_mb. x_bf16 uint16 [B,H,H,Cin], w1 f32 [Cexp,3,3,Cin],
    w2 f32 [Cout2,Cexp]. Returns (out uint16 [B,H/stride,H/stride,Cout2], ms_per_launch). f16: x / res / out hold fp16 bits."""
    x = np.ascontiguousarray(x_bf16, dtype=np.uint16)
    B, H, _, Cin = x.shape
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    w1, w2 = f(w1), f(w2)
    Cexp, Cout2 = w1.shape[0], w2.shape[0]
    out = np.empty((B, H // stride, H // stride, Cout2), np.uint16)
    r = None if res_bf16 is None else np.ascontiguousarray(res_bf16, dtype=np.uint16)
    ms = C.c_float()
    _lib.check(_lib.lib().isb_debug_fused_mb(device, _ptr(x), _ptr(w1), _ptr(f(scale1)), _ptr(f(shift1)), _ptr(w2),
                                             _ptr(f(scale2)), _ptr(f(shift2)), _ptr(r), B, H, Cin, Cexp, Cout2, stride | (0x100 if f16 else 0) | (0x200 if stamps else 0) | (0x400 if lds_e else 0), iters,
                                             _ptr(out), C.byref(ms)), "isb_debug_fused_mb")
    return out, ms.value


def gemm_f32_debug(A, W, bias=None, a_bias=None, a_add=None, act=0, a_act=0, splits=1, a_offset=0, iters=1, device=0):
    """The exact-f32 Linear kernel through isb_debug_gemm_f32. A f32 [M,K] or [a_parts,M,K], W f32 [N,K].
    Returns (C f32 [M,N], ms_per_launch)."""
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    A, W, bias, a_bias, a_add = f(A), f(W), f(bias), f(a_bias), f(a_add)
    if A.ndim == 2:
        A = A[None]
    a_parts, M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.float32)
    ms = C.c_float()
    p = lambda a: None if a is None else _ptr(a)
    _lib.check(_lib.lib().isb_debug_gemm_f32(device, _ptr(A), _ptr(W), p(bias), p(a_bias), p(a_add), M, N, K, a_parts, a_act,
                                             1 if a_add is None else a_add.shape[0], act, splits, a_offset, iters, _ptr(out),
                                             C.byref(ms)), "isb_debug_gemm_f32")
    return out, ms.value


def mbfront_debug(x16, w1, scale1, shift1, dww, dwscale, dwshift, f16=False, form=2, iters=1, device=0):
    """The front half of a stride-1 MBConv block through isb_debug_mbfront: x16 uint16 [B,hw,hw,cin] (hw 16: cin 192 / 224; hw 8: cin 384,
    2304 expanded channels), w1 f32 [cexp,cin], dww f32 [cexp,3,3]. form 0 = two launches, 1 = the fused kernel of rounds 4 / 5, 2 =
    producer / consumer waves. Returns (D uint16 [B,hw,hw,cexp], pooled f32 [B,cexp], ms_per_launch)."""
    x = np.ascontiguousarray(x16, dtype=np.uint16)
    B, hw, cin = x.shape[0], x.shape[1], x.shape[3]
    w1 = np.ascontiguousarray(w1, dtype=np.float32)
    cexp = w1.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (scale1, shift1, dww, dwscale, dwshift)]
    d = np.empty((B, hw, hw, cexp), np.uint16)
    pooled = np.empty((B, cexp), np.float32)
    ms = C.c_float()
    _lib.check(_lib.lib().isb_debug_mbfront(device, hw, _ptr(x), _ptr(w1), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]), _ptr(arrs[4]),
                                            B, cin, cexp, int(bool(f16)), int(form), int(iters), _ptr(d), _ptr(pooled), C.byref(ms)),
               "isb_debug_mbfront")
    return d, pooled, float(ms.value)




def se_fcs_debug(pooled, w1, b1, w2t, b2, iters=1, device=0):
    """The squeeze-excite FCs of a batch through isb_debug_se_fcs: pooled f32 [B,C], w1 [cse,C], b1 [cse], w2t [cse,C], b2 [C].
    Returns (gate f32 [B,C], ms per pair of launches)."""
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (pooled, w1, b1, w2t, b2)]
    B, Cc = arrs[0].shape
    cse = arrs[1].shape[0]
    gate = np.empty((B, Cc), np.float32)
    ms = C.c_float()
    _lib.check(_lib.lib().isb_debug_se_fcs(device, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]), _ptr(arrs[4]), B, Cc, cse,
                                           int(iters), _ptr(gate), C.byref(ms)), "isb_debug_se_fcs")
    return gate, float(ms.value)


def dwconv_debug(x_bf16, w, scale, shift, stride=1, iters=1, device=0, in_f16=False, out_f16=False, general=False):
    """Depthwise 3x3 + SiLU + SE mean through isb_debug_dwconv. x_bf16 uint16 [B,H,H,C], w f32 [C,3,3].
    Returns (out uint16 [B,H/stride,H/stride,C], pooled f32 [B,C], ms_per_launch). in_f16: x and the taps are fp16;
    out_f16: out is fp16 (DwArgs.in_f16 / out_f16); general (DwArgs.general, stride-1 8 x 8 / 16 x 16 maps): 0 / 2 = the LDS-map kernel
    with v_dot2 taps, 1 / True = the general kernel, 3 = the taps on the matrix pipe (the fused 8 x 8 front's arithmetic)."""
    x = np.ascontiguousarray(x_bf16, dtype=np.uint16)
    B, H, _, Cc = x.shape
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty((B, H // stride, H // stride, Cc), np.uint16)
    pooled = np.empty((B, Cc), np.float32)
    ms = C.c_float()
    _lib.check(_lib.lib().isb_debug_dwconv(device, _ptr(x), _ptr(f(w)), _ptr(f(scale)), _ptr(f(shift)), B, H, Cc,
                                           stride | (0x100 if in_f16 else 0) | (0x200 if out_f16 else 0) | {0: 0, 1: 0x400, 2: 0x800, 3: 0xc00}[int(general)],
                                           iters, _ptr(out), _ptr(pooled), C.byref(ms)), "isb_debug_dwconv")
    return out, pooled, ms.value


def dwconv_fc1_debug(x_bf16, w, scale, shift, se_w1, stride=1, device=0, in_f16=False, out_f16=False, general=False):
    """dwconv_debug with the squeeze-excite FC1 folded into the launch (isb_debug_dwconv_fc1). se_w1 f32 [cse,C].
    Returns (out, pooled, parts f32 [slabs,B,cse]): parts[s] = the slab's share of pooled @ se_w1.T."""
    x = np.ascontiguousarray(x_bf16, dtype=np.uint16)
    B, H, _, Cc = x.shape
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    se_w1 = f(se_w1)
    cse = se_w1.shape[0]
    out = np.empty((B, H // stride, H // stride, Cc), np.uint16)
    pooled = np.empty((B, Cc), np.float32)
    parts = np.zeros((32, B, cse), np.float32)
    ms = C.c_float()
    n = C.c_int32()
    _lib.check(_lib.lib().isb_debug_dwconv_fc1(device, _ptr(x), _ptr(f(w)), _ptr(f(scale)), _ptr(f(shift)), B, H, Cc,
                                               stride | (0x100 if in_f16 else 0) | (0x200 if out_f16 else 0) | (0x400 if general else 0),
                                               1, _ptr(out), _ptr(pooled), C.byref(ms), _ptr(se_w1), cse, _ptr(parts), C.byref(n)),
               "isb_debug_dwconv_fc1")
    return out, pooled, parts[:n.value]


def get_augmentations(num_aug: int, rot_aug_linspace_noend: bool = True):
    """Host-side mirror of the reference's test-time augmentation tables (modules/hpe/utils/misc.py:312-327): returns
    (should_flip bool[n], rotflip f64[n,3,3], gammas[n], scales[n]). Rotations about the optical axis spread over
    +-25 degrees (the end point left out by default), zoom 0.8..1.0 then 1.0..1.1, every other augmentation around the
    middle one mirrored in x. Same numpy calls, hence the same dtypes and roundings, as the reference's."""
    gammas = np.linspace(0.6, 1.0, num_aug)
    half_range = np.float32(np.deg2rad(25))
    n_angles = num_aug + 1 if rot_aug_linspace_noend else num_aug
    angles = np.linspace(-half_range, half_range, n_angles)[:num_aug]
    scales = np.concatenate([np.linspace(0.8, 1.0, (num_aug + 1) // 2)[:-1],
                             np.linspace(1.0, 1.1, num_aug - num_aug // 2)], axis=0)
    should_flip = (np.arange(num_aug) - num_aug // 2) % 2 != 0
    mirror_x = np.array([[-1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)
    maybe_mirror = np.where(should_flip[:, None, None], mirror_x, np.eye(3))
    a = -angles
    s, c, z, o = np.sin(a), np.cos(a), np.zeros_like(a), np.ones_like(a)
    rot = np.stack([np.stack([c, -s, z], axis=-1), np.stack([s, c, z], axis=-1), np.stack([z, z, o], axis=-1)], axis=-2)
    return should_flip, maybe_mirror @ rot, gammas, scales


def pose_windows(joints, seq_len: int):
    """joints torch CUDA f32 [n_cam, n_frames, J, 3] -> windows [n_cam*(n_frames-L+1), L, 3J]
    (root-centred on joint 0, main.py:103; window assembly, ar.py:42-50)."""
    import torch
    j = joints.contiguous().float()
    n_cam, n_frames, J, _ = j.shape
    nw = n_frames - seq_len + 1
    out = torch.empty((n_cam * nw, seq_len, 3 * J), dtype=torch.float32, device=j.device)
    stream = torch.cuda.current_stream(j.device).cuda_stream
    _lib.check(_lib.lib().isb_pose_windows(j.data_ptr(), n_cam, n_frames, J, seq_len, out.data_ptr(), C.c_void_p(stream)),
               "isb_pose_windows")
    return out


def pose_distance(joints):
    """joints torch CUDA f32 [..., J, 3] (absolute) -> f32 [...]: the frame's ``distance`` element (main.py:102)."""
    import torch
    j = joints.contiguous().float()
    J = j.shape[-2]
    n = j.numel() // (J * 3)
    out = torch.empty(j.shape[:-2], dtype=torch.float32, device=j.device)
    stream = torch.cuda.current_stream(j.device).cuda_stream
    _lib.check(_lib.lib().isb_pose_distance(j.data_ptr(), n, J, out.data_ptr(), C.c_void_p(stream)), "isb_pose_distance")
    return out


def load_joint_assets(expand_path: str, skeleton_types_path: str, skeleton: Optional[str]):
    expand = np.load(expand_path)
    with open(skeleton_types_path) as f:
        st = json.load(f)
    if skeleton is None:
        return expand, None, None
    return expand, st[skeleton]["indices"], [tuple(e) for e in st[skeleton]["edges"]]
