"""Drop-in for the reference's ``modules/ar/ar.py::ActionRecognizer`` (lines 11-96) on top of
libisbfsar_hip.so. Same constructor, methods, return values and public attributes, so the
reference's per-frame loop (``main.py:111``: ``self.ar.inference(ar_input)``), ``learn_command``
(``main.py:318``), ``forget_command`` (``main.py:207``) and ``save/load/debug``
(``main.py:213-226,321-333``, which read/replace ``support_set`` directly) run unmodified.

Host logic kept here (as in the reference): the sliding window of the last ``seq_len`` frames,
the class-name <-> index map, the class softmax and the result dict. Everything numeric
(MLP, tuple cross-attention, discriminator) happens in the HIP library; there is no CPU path.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from ...engine import ArEngine
from ...weights import state_from_torch, unpack_blob


def _to_numpy(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def _to_tensor(x):
    """support_set entries are torch tensors in the reference (ar.py:95) and main.py calls
    ``.detach().cpu().numpy()`` on them (main.py:216,226); keep that type when torch is there."""
    try:
        import torch
        return torch.from_numpy(np.array(x, dtype=np.float32, copy=True))
    except ImportError:  # pragma: no cover
        return np.array(x, dtype=np.float32, copy=True)


def _fingerprint(x):
    if x is None:
        return None
    a = np.ascontiguousarray(_to_numpy(x))
    try:
        import xxhash
        h = xxhash.xxh3_64_intdigest(a.data)
    except ImportError:  # pragma: no cover
        import zlib
        h = zlib.crc32(a.data)
    return (a.shape, h)


def _load_state(args):
    w = getattr(args, "weights", None)
    if w is not None:
        return w if not isinstance(w, (bytes, bytearray)) else unpack_blob(w)
    path = args.final_ckpt_path
    if path.endswith(".isbw"):
        with open(path, "rb") as f:
            return unpack_blob(f.read())
    import torch  # the reference's checkpoint format (ar.py:17-19)
    sd = torch.load(path, map_location="cpu")["model_state_dict"]
    return state_from_torch(sd)


class ActionRecognizer:
    def __init__(self, args, add_hook=False):
        self.input_type = args.input_type
        if self.input_type != "skeleton":
            raise NotImplementedError("the HIP build covers input_type='skeleton' (reference default, "
                                      "utils/params.py:4); rgb/hybrid are out of scope")
        self.device = args.device
        self.seq_len = args.seq_len
        self.way = args.way
        self.n_joints = args.n_joints
        self.ar = ArEngine(args.seq_len, args.n_joints, args.way,
                           device=getattr(args, "device_index", 0),
                           precision=getattr(args, "precision", "bf16"),
                           max_batch=getattr(args, "max_batch", 1024))
        self.ar.load_weights(_load_state(args))

        self.support_set = OrderedDict()
        self.requires_focus = {}
        self.previous_frames = []
        self._installed = None   # signature of the support set cached on the device

    # ------------------------------------------------------------------------------------
    def _signature(self):
        """Content fingerprint of the support set (class order, names, and the bytes of every poses / features array):
        what is on the device is keyed on WHAT the set holds, not on object identity -- in-place edits of
        ``support_set[c]["poses"]``, a replaced dict (main.py:323) and recycled object ids are all seen.
        xxh3 runs at >10 GB/s: ~20 us for the reference's 5 x [16,90] set, ~0.5 ms for 120 x [30,366]."""
        return tuple((k, _fingerprint(v.get("poses")), _fingerprint(v.get("features"))) for k, v in self.support_set.items())

    def _sync_support(self):
        """(Re)install the device-side support cache when ``support_set`` changed (train/remove/
        ``main.py:323`` load). Mirrors ar.py:56-74: use cached features when every class has them,
        else compute them from poses and cache them back."""
        sig = self._signature()
        if sig == self._installed:
            return
        names = list(self.support_set.keys())
        if len(names) > self.way:
            raise ValueError(f"{len(names)} classes exceed way={self.way}")
        if all("features" in self.support_set[c] for c in names):              # ar.py:56-61
            feats = np.stack([_to_numpy(self.support_set[c]["features"]) for c in names])
            self.ar.set_support(features=feats)
        else:                                                                  # ar.py:62-67
            poses = np.stack([_to_numpy(self.support_set[c]["poses"]) for c in names])
            self.ar.set_support(poses=poses)
            feats = self.ar.support_features()
            for i, c in enumerate(names):                                      # ar.py:72-74
                self.support_set[c]["features"] = _to_tensor(feats[i])
        self._installed = self._signature()

    # ------------------------------------------------------------------------------------
    def inference(self, data):
        """data: {"sk": array[3J]} -> (OrderedDict name->prob, is_true ndarray (1,) | 0, requires_focus)"""
        if data is None or len(data) == 0:                                     # ar.py:34-35
            return {}, 0, {}
        if len(self.support_set) == 0:                                         # ar.py:37-38
            return {}, 0, {}
        self.previous_frames.append({k: np.array(v, dtype=np.float32, copy=True) for k, v in data.items()})
        if len(self.previous_frames) < self.seq_len:                           # ar.py:43-44
            return {}, 0, {}
        elif len(self.previous_frames) == self.seq_len + 1:                    # ar.py:45-46
            self.previous_frames = self.previous_frames[1:]

        window = np.stack([f["sk"].reshape(-1) for f in self.previous_frames])[None]   # ar.py:49-50
        self._sync_support()
        logits, is_true, _ = self.ar.infer(window)                             # ar.py:69

        lg = logits[0].astype(np.float32)
        e = np.exp(lg - lg.max())
        few_shot_result = e / e.sum()                                          # ar.py:77
        open_set_result = is_true[:1].copy()                                   # ar.py:78, shape (1,)
        results = {}
        names = list(self.support_set.keys())
        for k in range(len(names)):                                            # ar.py:81-83
            results[names[k]] = few_shot_result[k]
        return results, open_set_result, self.requires_focus

    def remove(self, flag):                                                    # ar.py:86-92
        if flag in self.support_set.keys():
            self.support_set.pop(flag)
            self.requires_focus.pop(flag)
            return True
        else:
            return False

    def train(self, inp):                                                      # ar.py:94-96
        self.support_set[inp['flag']] = {c: _to_tensor(inp['data'][c]) for c in inp['data'].keys()}
        self.requires_focus[inp['flag']] = inp['requires_focus']
