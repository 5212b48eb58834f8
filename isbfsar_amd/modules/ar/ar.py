"""Drop-in for the reference's ``modules/ar/ar.py::ActionRecognizer`` (lines 11-96) on top of
libisbfsar_hip.so. Same constructor, methods, return values and public attributes, so the
reference's per-frame loop (``main.py:111``: ``self.ar.inference(ar_input)``), ``learn_command``
(``main.py:318``), ``forget_command`` (``main.py:207``) and ``save/load/debug``
(``main.py:213-226,321-333``, which read/replace ``support_set`` directly) run unmodified.

Host logic kept here (as in the reference): the sliding window of the last ``seq_len`` frames,
the class-name <-> index map, the class softmax and the result dict. Everything numeric
(MLP, tuple cross-attention, discriminator) happens in the HIP library; there is no CPU path.

input_type "skeleton" (the reference default, utils/params.py:4) and "hybrid" (utils/params.py:81; model.py:270-277,
296-316): in hybrid mode a frame is ``{"rgb": float[3,224,224], "sk": float[3J]}`` (main.py:85-105), a support class
``{"imgs": [L,3,224,224], "poses": [L,3J]}`` (ar.py:64-67); the ResNet-50 trunk runs in its own engine (RgbEngine) and its
[L,2048] features feed PostResNet inside the AR engine. "rgb" alone is refused: the reference sizes its transformer for
1000-wide features (utils/params.py:81) while its model produces 256 (model.py:274-277), so that mode cannot run there either.
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
    return state_from_torch(sd, hybrid=getattr(args, "input_type", "skeleton") == "hybrid")


def _load_rgb_state(args):
    """the ResNet-50 trunk's weights: args.rgb_weights (mapping / ISBW bytes), the `features_extractor.rgb.*` tensors of the
    reference's hybrid checkpoint (ar.py:17-19), or -- no pretrained torchvision weights here -- synthetic ones"""
    from ... import resnet50
    w = getattr(args, "rgb_weights", None)
    if w is not None:
        return w if not isinstance(w, (bytes, bytearray)) else unpack_blob(w)
    path = getattr(args, "final_ckpt_path", None)
    if path and not path.endswith(".isbw") and getattr(args, "weights", None) is None:
        import os
        if os.path.exists(path):
            import torch
            return resnet50.state_from_torch(torch.load(path, map_location="cpu")["model_state_dict"])
    return resnet50.make_state(getattr(args, "weights_seed", 0))


class ActionRecognizer:
    def __init__(self, args, add_hook=False):
        self.input_type = args.input_type
        if self.input_type not in ("skeleton", "hybrid"):
            raise NotImplementedError("input_type 'rgb': the reference's own configuration of it is inconsistent (transformer sized "
                                      "for 1000-wide features, utils/params.py:81; 256-wide features produced, model.py:274-277); "
                                      "'skeleton' and 'hybrid' are built")
        self.device = args.device
        self.seq_len = args.seq_len
        self.way = args.way
        self.n_joints = args.n_joints if args.input_type == "skeleton" else getattr(args, "n_joints", 0)   # ar.py:28 sets 0 for non-skeleton types
        n_j = args.n_joints
        self.ar = ArEngine(args.seq_len, n_j, args.way,
                           device=getattr(args, "device_index", 0),
                           precision=getattr(args, "precision", "default"),
                           max_batch=getattr(args, "max_batch", 1024), input_type=self.input_type)
        self.ar.load_weights(_load_state(args))
        self.rgb = None
        if self.input_type == "hybrid":
            from ...rgb_engine import RgbEngine
            self.rgb = RgbEngine(device=getattr(args, "device_index", 0), max_batch=max(64, args.seq_len))
            self.rgb.load_weights(_load_rgb_state(args))

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
        # (a class with cached features is used through them alone, ar.py:56-61: its raw data -- 9.6 MB of images per class in
        # hybrid mode -- is not hashed again on every frame)
        return tuple((k, _fingerprint(v.get("features"))) if "features" in v else
                     (k, _fingerprint(v.get("poses")), _fingerprint(v.get("imgs"))) for k, v in self.support_set.items())

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
            if self.rgb is not None:                                           # "imgs" [L,3,224,224] per class -> trunk features
                imgs = np.stack([_to_numpy(self.support_set[c]["imgs"]) for c in names])
                trunk = self.rgb.forward(imgs.reshape(-1, 3, 224, 224)).reshape(len(names), self.seq_len, 2048)
                self.ar.set_support(poses=poses, trunk=trunk)
            else:
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
        if self.rgb is not None:
            # the trunk features of a frame do not change while it slides through the window: computed once, when it arrives
            f = self.previous_frames[-1]
            if "trunk" not in f:
                for g in self.previous_frames:
                    if "trunk" not in g:
                        g["trunk"] = self.rgb.forward(g["rgb"].reshape(1, 3, 224, 224))[0]
            trunk = np.stack([g["trunk"] for g in self.previous_frames])[None]
            logits, is_true, _ = self.ar.infer(window, trunk=trunk)
        else:
            logits, is_true, _ = self.ar.infer(window)                         # ar.py:69

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
