"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in numpy, of the reference's skeleton action-recognition path:

    TRXOS.forward                    /root/reference/modules/ar/utils/model.py:291-328
      MLP                            model.py:164-180
      PositionalEncoding             model.py:12-28  (table length int(1.5*L), model.py:38-39)
      TemporalCrossTransformer       model.py:59-143 (tuples = combinations(range(L), 2), :52-54)
      Discriminator                  model.py:183-204 (sized at :283-285)
    ActionRecognizer state machine   /root/reference/modules/ar/ar.py:30-96

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product path (``isbfsar_amd``) never does. Parity status: PINNED -- checked in
``tests/test_oracle_golden.py`` against vectors produced by importing the reference's own
``TRXOS`` in the build container (``oracle/gen_golden.py``), committed under ``tests/golden``.

The arithmetic follows the reference statement by statement (tuples are materialised and the
512-wide k/v Linear is applied as written) so that it can be read next to model.py; the
``dtype`` argument lets tests run it in float64 as a higher-precision yardstick.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from itertools import combinations
from typing import Dict, Mapping, Optional

import numpy as np


def _linear(x, w, b):
    return x @ w.T + b


def positional_table(max_len: int, d_model: int, scale: float = 0.1, dtype=np.float32) -> np.ndarray:
    """model.py:17-23 -- computed in float32 like torch does (arange/exp/sin in f32)."""
    pe = np.zeros((max_len, d_model), np.float32)
    position = np.arange(0, max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d_model)))
    pe[:, 0::2] = np.sin(position * div_term) * np.float32(scale)
    pe[:, 1::2] = np.cos(position * div_term) * np.float32(scale)
    return pe.astype(dtype)


class TRXOSOracle:
    """Restatement of ``TRXOS`` restricted to ``input_type == 'skeleton'``, ``model == 'DISC'``."""

    def __init__(self, state: Mapping[str, np.ndarray], seq_len: int, n_joints: int,
                 d_in: int = 256, d_out: int = 128, dtype=np.float32):
        self.L, self.J, self.d_in, self.d_out = seq_len, n_joints, d_in, d_out
        self.dtype = dtype
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in state.items()}
        self.tuples = list(combinations(range(seq_len), 2))          # model.py:52-54
        self.T = len(self.tuples)
        self.idx0 = np.array([t[0] for t in self.tuples])
        self.idx1 = np.array([t[1] for t in self.tuples])
        self.pe = positional_table(int(seq_len * 1.5), d_in, dtype=dtype)  # model.py:38-39

    # -- model.py:164-180
    def mlp(self, x: np.ndarray) -> np.ndarray:
        w = self.w
        h = np.maximum(_linear(x.astype(self.dtype), w["features_extractor.sk.fc1.weight"],
                               w["features_extractor.sk.fc1.bias"]), 0)
        return np.maximum(_linear(h, w["features_extractor.sk.fc2.weight"],
                                  w["features_extractor.sk.fc2.bias"]), 0)

    # -- model.py:207-216 (PostResNet) + model.py:296-303 / 309-316 (feature concat, RGB first): input_type "hybrid"
    def post_resnet(self, trunk: np.ndarray) -> np.ndarray:
        """trunk [..., 2048] = output of the ResNet-50 trunk (global average pool) -> [..., 256]"""
        w = self.w
        return _linear(np.maximum(trunk.astype(self.dtype), 0), w["post_resnet.l1.weight"], w["post_resnet.l1.bias"])

    def features(self, poses: np.ndarray, trunk: Optional[np.ndarray] = None) -> np.ndarray:
        """skeleton: MLP(poses) [.., L, 256]; hybrid: [PostResNet(trunk) | MLP(poses)] [.., L, 512]"""
        sk = self.mlp(poses)
        if trunk is None:
            return sk
        return np.concatenate([self.post_resnet(trunk), sk], axis=-1)

    def forward_hybrid(self, ss_poses, ss_trunk, n_classes: int, q_poses, q_trunk, ss_features=None) -> Dict[str, np.ndarray]:
        """TRXOS.forward with query_data = {"rgb", "sk"} (model.py:291-328), the RGB side given as trunk features [.., L, 2048]"""
        assert self.d_in == 512
        q_feats = self.features(q_poses, q_trunk)
        if ss_features is None:
            ss_features = self.features(ss_poses, ss_trunk)
        logits, diffs = self.cross_transformer(np.asarray(ss_features, self.dtype), n_classes, q_feats)
        chosen = np.argmax(logits, axis=1)
        is_true = self.discriminator(diffs[np.arange(q_poses.shape[0]), chosen])
        return {"logits": logits, "is_true": is_true, "support_features": ss_features, "query_features": q_feats, "chosen": chosen}

    # -- model.py:65-84
    def tuples_kv(self, feats: np.ndarray):
        """feats [..., L, 256] (pre-PE) -> K [..., T, 128] (LayerNorm'ed), V [..., T, 128]."""
        w = self.w
        x = feats + self.pe[: feats.shape[-2]]                        # model.py:26
        t = np.concatenate([x[..., self.idx0, :], x[..., self.idx1, :]], axis=-1)  # model.py:69-72
        k = _linear(t, w["transformers.0.k_linear.weight"], w["transformers.0.k_linear.bias"])
        v = _linear(t, w["transformers.0.v_linear.weight"], w["transformers.0.v_linear.bias"])
        mu = k.mean(-1, keepdims=True)
        var = ((k - mu) ** 2).mean(-1, keepdims=True)                 # biased, like nn.LayerNorm
        k = (k - mu) / np.sqrt(var + self.dtype(1e-5)) * w["transformers.0.norm_k.weight"] \
            + w["transformers.0.norm_k.bias"]
        return k, v

    # -- model.py:95-143
    def cross_transformer(self, ss_feats: np.ndarray, n_classes: int, q_feats: np.ndarray):
        """ss_feats [way_pad, L, 256] shared by every query; q_feats [B, L, 256].
        Returns logits [B, n], diffs [B, n, T, 128]."""
        kq, vq = self.tuples_kv(q_feats)                              # [B,T,128]
        ks, vs = self.tuples_kv(ss_feats[:n_classes])                 # [n,T,128]
        scale = self.dtype(1.0 / math.sqrt(self.d_out))
        B = q_feats.shape[0]
        logits = np.empty((B, n_classes), self.dtype)
        diffs = np.empty((B, n_classes, self.T, self.d_out), self.dtype)
        for c in range(n_classes):                                    # model.py:95
            s = (kq @ ks[c].T) * scale                                # [B,T,T]  model.py:101-102
            s = s - s.max(axis=-2, keepdims=True)                     # softmax over dim=-2 (model.py:49,109)
            e = np.exp(s)
            a = e / e.sum(axis=-2, keepdims=True)
            proto = a @ vs[c]                                         # model.py:127
            diff = vq - proto                                         # model.py:132
            diffs[:, c] = diff
            logits[:, c] = -(diff.astype(self.dtype) ** 2).sum(axis=(-2, -1)) / self.dtype(self.T)  # :133-137
        return logits, diffs

    # -- model.py:194-204
    def discriminator(self, x: np.ndarray) -> np.ndarray:
        w = self.w
        y = _linear(x, w["discriminator.dimensionality_reduction.weight"],
                    w["discriminator.dimensionality_reduction.bias"])  # [B,T,L]
        y = y.reshape(x.shape[0], -1)
        y = np.maximum(_linear(y, w["discriminator.fc1.weight"], w["discriminator.fc1.bias"]), 0)
        y = np.maximum(_linear(y, w["discriminator.fc2.weight"], w["discriminator.fc2.bias"]), 0)
        y = _linear(y, w["discriminator.fc3.weight"], w["discriminator.fc3.bias"])
        return 1.0 / (1.0 + np.exp(-y))

    # -- model.py:291-328
    def forward(self, ss_poses: Optional[np.ndarray], n_classes: int, q_poses: np.ndarray,
                ss_features: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        """ss_poses [n, L, 3J] (or None with ss_features [n, L, 256]); q_poses [B, L, 3J]."""
        q_feats = self.mlp(q_poses)
        if ss_features is None:
            ss_features = self.mlp(ss_poses)
        logits, diffs = self.cross_transformer(np.asarray(ss_features, self.dtype), n_classes, q_feats)
        chosen = np.argmax(logits, axis=1)                            # model.py:323
        feature = diffs[np.arange(q_poses.shape[0]), chosen]          # model.py:324
        is_true = self.discriminator(feature)                         # model.py:325
        return {"logits": logits, "is_true": is_true, "support_features": ss_features,
                "query_features": q_feats, "chosen": chosen}


def softmax1d(x: np.ndarray) -> np.ndarray:
    e = np.exp(x - x.max())
    return e / e.sum()


class ActionRecognizerOracle:
    """Restatement of ``modules/ar/ar.py:11-96`` (sliding window, feature cache, class softmax)
    on top of ``TRXOSOracle``. torch/cuda plumbing is dropped; the control flow is kept."""

    def __init__(self, net: TRXOSOracle, way: int):
        self.ar = net
        self.support_set: "OrderedDict[str, dict]" = OrderedDict()
        self.requires_focus: dict = {}
        self.previous_frames: list = []
        self.seq_len = net.L
        self.way = way
        self.n_joints = net.J

    def inference(self, data):
        if data is None or len(data) == 0:                            # ar.py:34-35
            return {}, 0, {}
        if len(self.support_set) == 0:                                # ar.py:37-38
            return {}, 0, {}
        self.previous_frames.append(np.asarray(data["sk"], np.float32).copy())  # ar.py:41-42
        if len(self.previous_frames) < self.seq_len:                  # ar.py:43-44
            return {}, 0, {}
        elif len(self.previous_frames) == self.seq_len + 1:           # ar.py:45-46
            self.previous_frames = self.previous_frames[1:]
        q = np.stack(self.previous_frames)[None]                      # ar.py:49-50
        names = list(self.support_set.keys())
        n = len(names)
        if all("features" in self.support_set[c] for c in names):     # ar.py:56-61
            ss_f = np.stack([self.support_set[c]["features"] for c in names])
            out = self.ar.forward(None, n, q, ss_features=ss_f)
        else:                                                         # ar.py:62-67
            ss = np.stack([self.support_set[c]["poses"] for c in names])
            out = self.ar.forward(ss, n, q)
            for i, s in enumerate(names):                             # ar.py:72-74
                self.support_set[s]["features"] = out["support_features"][i]
        few_shot = softmax1d(out["logits"][0])                        # ar.py:77
        open_set = out["is_true"][0]                                  # ar.py:78
        results = {names[k]: few_shot[k] for k in range(n)}           # ar.py:81-83
        return results, open_set, self.requires_focus

    def remove(self, flag):                                           # ar.py:86-92
        if flag in self.support_set:
            self.support_set.pop(flag)
            self.requires_focus.pop(flag)
            return True
        return False

    def train(self, inp):                                             # ar.py:94-96
        self.support_set[inp["flag"]] = {c: np.asarray(inp["data"][c], np.float32) for c in inp["data"]}
        self.requires_focus[inp["flag"]] = inp["requires_focus"]
