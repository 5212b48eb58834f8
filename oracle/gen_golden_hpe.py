"""Golden vectors for the pose path around the backbone -- runs ONLY in the build container.

Executes the reference's own code (nothing is copied):
  * ``modules/hpe/utils/misc.py`` imported unmodified,
  * ``modules/hpe/hpe.py::HumanPoseEstimator.estimate`` with ``Runner`` replaced by a fake that
    returns (i) a hand-made YOLO tensor with one person box, (ii) the output of the reference's
    own torch ``ImageTransformer`` (``modules/hpe/setup/6_create_image_transformation_onnx.py``)
    with ``.cuda()`` patched to identity, (iii) zeros for the backbone and (iv) seeded head logits.
    tensorrt / pycuda / loguru / cv2 are stubbed: they are not installed and only matter before
    the detector, which is out of scope.
Also converts the two joint assets the hot path consumes (``assets/32_to_122.npy``,
``assets/skeleton_types.pkl`` -- read through an allow-list unpickler) into
``isbfsar_amd/assets`` as .npy / .json.
"""
from __future__ import annotations

import hashlib
import importlib.util
import io
import json
import os
import pickle
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BBOXES = [(100, 200, 100, 200), (0, 359, 112, 478), (200, 440, 60, 420), (192, 448, 48, 432)]  # x1,x2,y1,y2


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class _SafeUnpickler(pickle.Unpickler):
    _ALLOWED = {("numpy.core.multiarray", "scalar"), ("numpy", "dtype"), ("numpy._core.multiarray", "scalar"),
                ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"blocked {module}.{name}")


def safe_load(f):
    return _SafeUnpickler(f).load()


def convert_assets():
    dst = os.path.join(ROOT, "isbfsar_amd", "assets")
    os.makedirs(dst, exist_ok=True)
    w = np.load(os.path.join(REF, "assets", "32_to_122.npy"))
    np.save(os.path.join(dst, "32_to_122.npy"), w.astype(np.float32))
    with open(os.path.join(REF, "assets", "skeleton_types.pkl"), "rb") as f:
        st = safe_load(f)
    out = {}
    for k, v in st.items():
        out[k] = {"indices": [int(i) for i in v["indices"]],
                  "edges": [[int(a), int(b)] for a, b in v["edges"]]}
        for extra in v:
            if extra not in ("indices", "edges"):
                try:
                    out[k][extra] = json.loads(json.dumps(v[extra], default=lambda o: np.asarray(o).tolist()))
                except Exception:
                    pass
    with open(os.path.join(dst, "skeleton_types.json"), "w") as f:
        json.dump(out, f)
    print("assets:", {k: (len(v["indices"]), len(v["edges"])) for k, v in out.items()})


def _stub_modules():
    trt = types.ModuleType("tensorrt")
    pc = types.ModuleType("pycuda")
    pcd = types.ModuleType("pycuda.driver")
    pc.driver = pcd
    lg = types.ModuleType("loguru")
    lg.logger = types.SimpleNamespace(info=lambda *a, **k: None, success=lambda *a, **k: None)
    cv2 = types.ModuleType("cv2")
    cv2.INTER_AREA = 3
    cv2.COLOR_BGR2RGB = 4
    cv2.resize = lambda img, size, **kw: np.zeros((size[1], size[0], 3), np.uint8)
    cv2.cvtColor = lambda img, code: img
    mpl = types.ModuleType("utils.matplotlib_visualizer")
    mpl.MPLPosePrinter = object
    tq = types.ModuleType("tqdm")
    tq.tqdm = lambda x, **k: x
    sys.modules.update({"tensorrt": trt, "pycuda": pc, "pycuda.driver": pcd, "loguru": lg, "cv2": cv2,
                        "utils.matplotlib_visualizer": mpl, "tqdm": tq})


def _reference_image_transformer():
    import torch
    spec = importlib.util.spec_from_file_location(
        "ref_img_tf", os.path.join(REF, "modules/hpe/setup/6_create_image_transformation_onnx.py"))
    mod = importlib.util.module_from_spec(spec)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # its __init__ calls .cuda() (:17-20)
    try:
        spec.loader.exec_module(mod)
        tf = mod.ImageTransformer(1, 480, 640)
    finally:
        torch.Tensor.cuda = orig
    return tf


class FakeRunner:
    """Stands in for utils/tensorrt_runner.py::Runner (flat numpy in, list of flat numpy out)."""
    state = {}

    def __init__(self, engine_path):
        self.role = os.path.basename(engine_path).split(".")[0]

    def __call__(self, *args):
        import torch
        st = FakeRunner.state
        if self.role == "yolo":
            x1, x2, y1, y2 = st["bbox"]
            boxes = np.zeros((1, 4032, 1, 4), np.float32)
            confs = np.zeros((1, 4032, 80), np.float32)
            boxes[0, 0, 0] = [(x1 + 0.5) / 640, (y1 + 0.5) / 480, (x2 + 0.5) / 640, (y2 + 0.5) / 480]
            confs[0, 0, 0] = 0.9
            return [boxes.ravel(), confs.ravel()]
        if self.role.startswith("image_transformation"):
            frame, H = args
            st["H"] = np.array(H, copy=True)
            with torch.no_grad():
                out = st["tf"](torch.from_numpy(np.asarray(frame)).int(), torch.from_numpy(np.asarray(H, np.float32)))
            st["warp_int"] = out.numpy().copy()
            return [out.numpy().astype(np.int32).ravel()]
        if self.role.startswith("bbone"):
            st["bbone_in"] = np.array(args[0], copy=True)
            return [np.zeros(8 * 8 * 1280, np.float32)]
        if self.role.startswith("heads"):
            return [st["head_logits"].ravel()]
        raise AssertionError(self.role)


def reference_estimator(skeleton="smpl+head_30"):
    _stub_modules()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import modules.hpe.hpe as ref_hpe
        from utils.params import MetrabsTRTConfig, RealSenseIntrinsics
        ref_hpe.Runner = FakeRunner
        ref_hpe.pickle = types.SimpleNamespace(load=safe_load)    # allow-list unpickler for skeleton_types.pkl
        cfg = MetrabsTRTConfig()
        cfg.skeleton = skeleton
        est = ref_hpe.HumanPoseEstimator(cfg, RealSenseIntrinsics(), just_box=False)
    finally:
        os.chdir(cwd)
    return est


def head_logits_case(seed: int, kind: str) -> np.ndarray:
    rng = np.random.default_rng(4242 + seed)
    lg = rng.normal(0.0, 3.0, (1, 8, 8, 288)).astype(np.float32)
    if kind == "peaked":          # sharp heatmaps spread over the crop -> mixed in/out of FOV joints
        for j in range(32):
            hh, ww, dd = rng.integers(0, 8, 3)
            lg[0, hh, ww, j] += 12.0
            lg[0, hh, ww, 32 + dd * 32 + j] += 12.0
    elif kind == "corner":        # everything at the top-left corner -> out of FOV -> estimate() returns None
        lg[0, 0, 0, :] += 30.0
    return lg


def gen_all(out_dir: str):
    import torch
    convert_assets()
    tf = _reference_image_transformer()
    FakeRunner.state["tf"] = tf
    est30 = reference_estimator("smpl+head_30")
    est122 = reference_estimator(None)
    cases = []
    frames = {}
    for ci, (bbox, seed, kind, est, tag) in enumerate([
            (BBOXES[0], 0, "gauss", est30, "30"), (BBOXES[1], 1, "peaked", est30, "30"),
            (BBOXES[2], 2, "peaked", est30, "30"), (BBOXES[3], 3, "gauss", est122, "122"),
            (BBOXES[2], 4, "corner", est30, "30"), (BBOXES[3], 5, "peaked", est122, "122")]):
        frame = np.random.default_rng(seed).integers(0, 256, (480, 640, 3), dtype=np.uint8)
        st = FakeRunner.state
        st["bbox"] = bbox
        st["head_logits"] = head_logits_case(seed, kind)
        res = est.estimate(frame)
        rec = {"bbox": np.array(bbox, np.int32), "frame_seed": seed, "kind": kind, "skeleton": tag,
               "head_logits": st["head_logits"], "H": st["H"].astype(np.float32),
               "warp_digest": digest(st["warp_int"].astype(np.uint8)),
               "warp_patch": st["warp_int"][0, :32, :32].astype(np.uint8),
               "warp_center": st["warp_int"][0, 112:144, 112:144].astype(np.uint8),
               "bbone_in_digest": digest(st["bbone_in"].astype(np.float32)),
               "valid": res is not None}
        if res is not None:
            assert tuple(res["bbox"]) == tuple(bbox), (res["bbox"], bbox)
            rec["pose"] = np.asarray(res["pose"], np.float64)
            rec["n_edges"] = 0 if res["edges"] is None else len(res["edges"])
        cases.append(rec)
        print(f"hpe case {ci}: bbox={bbox} kind={kind} valid={res is not None}"
              + (f" pose[0]={res['pose'][0]}" if res is not None else ""))
    flat = {}
    for i, rec in enumerate(cases):
        for k, v in rec.items():
            flat[f"c{i}_{k}"] = v
    flat["n_cases"] = len(cases)
    # G4: homography known answers straight from misc.homography (f64) for every bbox
    from modules.hpe.utils.misc import homography as ref_homography
    for i, (x1, x2, y1, y2) in enumerate(BBOXES):
        new_K, R = ref_homography(x1, x2, y1, y2, est30.K, 256)
        flat[f"hom{i}_new_K"] = new_K
        flat[f"hom{i}_R"] = R
        flat[f"hom{i}_H"] = (est30.K @ np.linalg.inv(new_K @ R)).astype(np.float32)
    flat["bboxes"] = np.array(BBOXES, np.int32)
    # G8: detector post-processing (misc.postprocess_yolo_output + hpe.py:63-79) on seeded YOLO tensors
    from modules.hpe.utils.misc import postprocess_yolo_output as ref_post
    from isbfsar_amd import synth
    yb, yc = synth.yolo_outputs()
    sel = []
    for i in range(yb.shape[0]):
        box = ref_post(yb[i:i + 1], yc[i:i + 1], est30.yolo_thresh, est30.nms_thresh)[0]
        humans = [e for e in box if e[5] == 0]
        if not humans:
            sel.append([-1, -1, -1, -1])
            continue
        humans.sort(key=lambda x: x[4], reverse=True)
        hm = humans[0]
        x1 = int(hm[0] * 640) if int(hm[0] * 640) > 0 else 0
        y1 = int(hm[1] * 480) if int(hm[1] * 480) > 0 else 0
        x2 = int(hm[2] * 640) if int(hm[2] * 640) > 0 else 0
        y2 = int(hm[3] * 480) if int(hm[3] * 480) > 0 else 0
        sel.append([x1, x2, y1, y2])
    flat["yolo_sel"] = np.array(sel, np.int32)
    flat["yolo_boxes_digest"] = digest(yb)
    flat["yolo_confs_digest"] = digest(yc)
    print("yolo selections:", sel)
    np.savez_compressed(os.path.join(out_dir, "hpe_post.npz"), **flat)
    print("hpe goldens written")


def gen_tta(out_dir: str, num_aug: int = 5):
    """G9: test-time augmentation as far as the reference executes it. estimate() is run with num_aug = 5 (n_test = 5,
    ImageTransformer(5, 480, 640)); the FakeRunner records the H matrices and the warped crops it is handed
    (hpe.py:88-100). What follows in estimate() reshapes the head output to one sample (hpe.py:108) and is not defined
    for five crops: whatever it does with the fake one-sample logits is ignored here."""
    import torch
    spec = importlib.util.spec_from_file_location(
        "ref_img_tf5", os.path.join(REF, "modules/hpe/setup/6_create_image_transformation_onnx.py"))
    mod = importlib.util.module_from_spec(spec)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        spec.loader.exec_module(mod)
        tf5 = mod.ImageTransformer(num_aug, 480, 640)
    finally:
        torch.Tensor.cuda = orig
    FakeRunner.state["tf"] = tf5
    est = reference_estimator("smpl+head_30")
    est.num_aug = num_aug
    est.n_test = num_aug
    from modules.hpe.utils.misc import get_augmentations as ref_aug, homography as ref_homography
    flip, rotflip, gammas, scales = ref_aug(num_aug)
    flat = {"num_aug": num_aug, "aug_should_flip": flip, "aug_rotflip": rotflip, "aug_gammas": gammas, "aug_scales": scales,
            "aug_rotflip_dtype": str(rotflip.dtype), "aug_scales_dtype": str(scales.dtype)}
    boxes = [BBOXES[2], BBOXES[3], BBOXES[0]]
    for i, bbox in enumerate(boxes):
        frame = np.random.default_rng(100 + i).integers(0, 256, (480, 640, 3), dtype=np.uint8)
        st = FakeRunner.state
        st["bbox"] = bbox
        st["head_logits"] = head_logits_case(i, "gauss")
        st.pop("H", None)
        try:
            est.estimate(frame)
            after = "returned"
        except Exception as e:                      # the one-sample decode / reconstruction of five crops
            after = f"{type(e).__name__}: {e}"
        H = np.asarray(st["H"], np.float32).reshape(num_aug, 3, 3)
        wi = st["warp_int"].astype(np.uint8)        # [5,256,256,3]
        # the same quantities through the reference's functions (hpe.py:85-96 spelled out)
        new_K, homo_inv = ref_homography(*bbox, est.K, 256)
        nk = np.tile(new_K, (num_aug, 1, 1))
        for k in range(num_aug):
            nk[k, :2, :2] *= scales[k]
        hi = rotflip @ np.tile(homo_inv[0], (num_aug, 1, 1))
        H2 = (est.K @ np.linalg.inv(nk @ hi)).astype(np.float32)
        assert np.array_equal(H, H2), "estimate() and the spelled-out lines disagree"
        flat.update({f"t{i}_bbox": np.array(bbox, np.int32), f"t{i}_frame_seed": 100 + i, f"t{i}_H": H, f"t{i}_new_K": nk,
                     f"t{i}_homo_inv": hi, f"t{i}_after_warp": after,
                     f"t{i}_warp_digest": np.array([digest(wi[k]) for k in range(num_aug)]),
                     f"t{i}_warp_patch": wi[:, 96:160:2, 96:160:2].copy(),
                     f"t{i}_warp_nonzero": np.array([int((wi[k] != 0).any(axis=-1).sum()) for k in range(num_aug)]),
                     f"t{i}_bbone_in_digest": digest(np.asarray(st["bbone_in"], np.float32))})
        print(f"tta case {i}: bbox={bbox} H[0,0]={H[0, 0]} after warp: {after[:80]}")
    flat["n_cases"] = len(boxes)
    np.savez_compressed(os.path.join(out_dir, "hpe_tta.npz"), **flat)
    print("tta goldens written")


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "tta":
        gen_tta(os.path.join(ROOT, "tests", "golden"))
    else:
        gen_all(os.path.join(ROOT, "tests", "golden"))
        gen_tta(os.path.join(ROOT, "tests", "golden"))
