"""Weight container ("ISBW" blob) and the deterministic weight generator.

The reference ships no weights (``modules/ar/modules/raws/DISC.pth`` and the MetrABS
engines are git-ignored: reference ``.gitignore:9-24,45``), so parity is established on
deterministically generated tensors that can be loaded BOTH into the reference's own
``TRXOS`` (``modules/ar/utils/model.py:219``) and into the HIP library.

Blob layout (little endian), consumed by ``isb_ar_load_weights`` / ``isb_hpe_load_weights``
(``include/isbfsar.h``):

    char[4]  magic  = "ISBW"
    u32      version = 1
    u32      n_tensors
    u32      reserved
    n_tensors x { char[96] name (NUL padded); u32 ndim; u32 dims[4]; u32 pad; u64 offset; u64 nbytes }
    payload  fp32 tensors, each 64-byte aligned, offsets relative to blob start

The generator is counter based (splitmix64 over ``fnv1a(name) ^ seed + index``) so that any
tensor can be regenerated anywhere without torch's RNG or a fixture file.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np

MAGIC = b"ISBW"
VERSION = 1
_NAME_LEN = 96
_ENTRY = struct.Struct("<96sI4I4xQQ")
_HEADER = struct.Struct("<4sIII")

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (name, seed, index)."""
    base = np.uint64((fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
    z = _splitmix64(ctr)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(name: str, shape: Tuple[int, ...], lo: float, hi: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# AR (TRXOS skeleton branch) state dict. Key names/shapes follow the reference modules:
#   MLP                      modules/ar/utils/model.py:164-180
#   TemporalCrossTransformer modules/ar/utils/model.py:31-57
#   Discriminator            modules/ar/utils/model.py:183-192, sized at :283-285
# --------------------------------------------------------------------------------------
def ar_state_shapes(seq_len: int, n_joints: int, d_in: int = 256, d_out: int = 128, hybrid: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """hybrid (TRXConfig.input_type == "hybrid", utils/params.py:81): per-frame features = [PostResNet(rgb trunk) 256 | skeleton MLP 256],
    so the transformer's input width is 512 and post_resnet.l1 (model.py:207-216) is on the path."""
    L, J = seq_len, n_joints
    T = L * (L - 1) // 2
    if hybrid:
        d_in = 512
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if hybrid:
        s["post_resnet.l1.weight"] = (256, 2048)
        s["post_resnet.l1.bias"] = (256,)
    s["features_extractor.sk.fc1.weight"] = (6 * J, 3 * J)
    s["features_extractor.sk.fc1.bias"] = (6 * J,)
    s["features_extractor.sk.fc2.weight"] = (256, 6 * J)
    s["features_extractor.sk.fc2.bias"] = (256,)
    s["transformers.0.k_linear.weight"] = (d_out, 2 * d_in)
    s["transformers.0.k_linear.bias"] = (d_out,)
    s["transformers.0.v_linear.weight"] = (d_out, 2 * d_in)
    s["transformers.0.v_linear.bias"] = (d_out,)
    s["transformers.0.norm_k.weight"] = (d_out,)
    s["transformers.0.norm_k.bias"] = (d_out,)
    s["discriminator.dimensionality_reduction.weight"] = (L, d_out)
    s["discriminator.dimensionality_reduction.bias"] = (L,)
    s["discriminator.fc1.weight"] = (256, T * L)
    s["discriminator.fc1.bias"] = (256,)
    s["discriminator.fc2.weight"] = (64, 256)
    s["discriminator.fc2.bias"] = (64,)
    s["discriminator.fc3.weight"] = (1, 64)
    s["discriminator.fc3.bias"] = (1,)
    return s


def make_ar_state(seq_len: int, n_joints: int, seed: int = 0, gain: float = 1.0, disc_gain: float = 1.0,
                  norm_gain: float = 1.0, hybrid: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Deterministic TRXOS (skeleton) weights: U(-g/sqrt(fan_in), g/sqrt(fan_in)) like
    torch's Linear default; LayerNorm gamma in [0.8,1.2], beta in [-0.1,0.1] so that the
    affine part of ``norm_k`` (model.py:46) is exercised.

    ``disc_gain`` multiplies the four Discriminator weight matrices (model.py:186-191): with the default
    initialisation the open-set score sits in 0.50-0.51 whatever the input, with 6 it spans (0.05, 0.95) over
    the synthetic windows, so a wrong ``diff`` shows. ``norm_gain`` multiplies ``norm_k.weight`` (a trained
    LayerNorm gain: sharper tuple attention, model.py:46,101-109)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    shapes = ar_state_shapes(seq_len, n_joints, hybrid=hybrid)
    for name, shape in shapes.items():
        if name.endswith("norm_k.weight"):
            out[name] = uniform(name, shape, 0.8, 1.2, seed)
        elif name.endswith("norm_k.bias"):
            out[name] = uniform(name, shape, -0.1, 0.1, seed)
        else:
            # bias shares its Linear's fan_in (torch.nn.Linear.reset_parameters)
            wshape = shapes[name[:-len("bias")] + "weight"] if name.endswith(".bias") else shape
            fan_in = wshape[-1]
            b = gain / np.sqrt(float(fan_in))
            out[name] = uniform(name, shape, -b, b, seed)
    if disc_gain != 1.0:
        for name in ("discriminator.dimensionality_reduction.weight", "discriminator.fc1.weight",
                     "discriminator.fc2.weight", "discriminator.fc3.weight"):
            out[name] = (out[name] * np.float32(disc_gain)).astype(np.float32)
    if norm_gain != 1.0:
        out["transformers.0.norm_k.weight"] = (out["transformers.0.norm_k.weight"] * np.float32(norm_gain)).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------
# blob (de)serialisation
# --------------------------------------------------------------------------------------
def pack_blob(tensors: Mapping[str, np.ndarray]) -> bytes:
    names = list(tensors.keys())
    n = len(names)
    table_bytes = _HEADER.size + n * _ENTRY.size
    off = (table_bytes + 63) // 64 * 64
    entries = []
    payload = []
    for name in names:
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        if a.ndim > 4:
            raise ValueError(f"{name}: ndim {a.ndim} > 4")
        if len(name.encode()) >= _NAME_LEN:
            raise ValueError(f"tensor name too long: {name}")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        entries.append(_ENTRY.pack(name.encode(), a.ndim, *dims, off, a.nbytes))
        payload.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    buf = bytearray(off)
    _HEADER.pack_into(buf, 0, MAGIC, VERSION, n, 0)
    p = _HEADER.size
    for e in entries:
        buf[p:p + _ENTRY.size] = e
        p += _ENTRY.size
    for o, b in payload:
        buf[o:o + len(b)] = b
    return bytes(buf)


def unpack_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    magic, ver, n, _ = _HEADER.unpack_from(blob, 0)
    if magic != MAGIC or ver != VERSION:
        raise ValueError("not an ISBW v1 blob")
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    p = _HEADER.size
    for _ in range(n):
        name, ndim, d0, d1, d2, d3, off, nbytes = _ENTRY.unpack_from(blob, p)
        p += _ENTRY.size
        shape = (d0, d1, d2, d3)[:ndim]
        a = np.frombuffer(blob, dtype=np.float32, count=nbytes // 4, offset=off).reshape(shape)
        out[name.rstrip(b"\0").decode()] = a
    return out


def state_from_torch(state_dict: Mapping[str, "object"], keys: Iterable[str] | None = None, hybrid: bool = False) -> Dict[str, np.ndarray]:
    """Convert a torch ``state_dict`` (e.g. the reference's ``DISC.pth['model_state_dict']``,
    ``modules/ar/ar.py:17-19``) into the numpy mapping ``pack_blob`` takes.

    * ``.module`` infixes left by ``DataParallel`` are stripped exactly as ``ar.py:18`` does;
    * checkpoints written before the reference grew its RGB branch name the skeleton MLP
      ``features_extractor.fc1/fc2``; the reference migrates them once with
      ``utils/rename_torch_layers_and_parameters.py:11`` (``features_extractor`` -> ``features_extractor.sk``):
      the same rename is applied here to keys that do not carry the ``.sk`` level yet;
    * ``post_resnet.*`` (RGB branch, zero-filled by that script, lines 12-13) is not on the skeleton path and
      is dropped -- unless ``hybrid``: then ``post_resnet.l1.*`` is kept (it is on the hybrid path; the ResNet-50 trunk under
      ``features_extractor.rgb.*`` goes through ``resnet50.state_from_torch`` into the RGB engine's own blob)."""
    out = {}
    for k, v in state_dict.items():
        k2 = k.replace(".module", "")
        # the RGB branch is not on the skeleton path: drop it BEFORE the rename below would hide its prefix
        if k2.startswith("features_extractor.rgb.") or (k2.startswith("post_resnet.") and not hybrid):
            continue
        if k2.startswith("features_extractor.") and not k2.startswith("features_extractor.sk.") and not k2.startswith("post_resnet."):
            k2 = "features_extractor.sk." + k2[len("features_extractor."):]
        if keys is not None and k2 not in keys:
            continue
        out[k2] = np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
    return out
