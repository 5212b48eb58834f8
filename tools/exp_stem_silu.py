"""what the stem's SiLU arithmetic is worth on the hard ("signal") weight profile, statistically.
    python tools/exp_stem_silu.py [n_frames=64]
The fp32 oracle's poses are computed once (both weight profiles), then one child process per ISB_STEM_SILU value (0 = v_rcp, 1 = v_rcp +
one Newton step, 2 = IEEE division: the switch is read once per process) runs the default precision on the same frames and prints the
per-frame distribution of |absolute pose - fp32 definition| and the stem kernel's rate."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isbfsar_amd import effnetv2, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "--child" else 64
REFS = os.path.join(ROOT, "gpurun_out", "stem_silu_refs.npz")


def states():
    return {"default": effnetv2.make_state(0), "signal": effnetv2.make_state(0, "signal", head_gain=0.5)}


def K():
    from oracle import hpe_oracle as ho
    return ho.intrinsics_matrix(384.025146484375, 384.025146484375, 319.09661865234375, 237.75723266601562)


def parent():
    import torch
    from oracle import hpe_oracle as ho
    from oracle.effnetv2_oracle import EffNetV2LOracle
    from bench_workloads import usable_cores          # (the cgroup's share, not the host's core count: oversubscribed torch crawls)
    torch.set_num_threads(usable_cores())
    W = np.load(os.path.join(ROOT, "isbfsar_amd", "assets", "32_to_122.npy"))
    fr, bb = synth.frames(N, seed=0), synth.bboxes(N, seed=0)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], K())[2][0]) for b in range(N)])
    out = {}
    t0 = time.time()
    for prof, st in states().items():
        print(f"oracle, profile {prof} ...", flush=True)
        net = EffNetV2LOracle(st, "f32")
        lg = np.concatenate([net.head(net.backbone(crops[i:i + 16])) for i in range(0, N, 16)])
        poses = [ho.postprocess(lg[j:j + 1], *ho.crop_params(bb[j], K())[:2], W, None) for j in range(N)]
        out[prof + "_ok"] = np.array([p is not None for p in poses])
        out[prof + "_pose"] = np.stack([p if p is not None else np.zeros((122, 3)) for p in poses])
        out[prof + "_p3"] = ho.decode(lg)[1]
    os.makedirs(os.path.dirname(REFS), exist_ok=True)
    np.savez(REFS, n=N, **out)
    print(f"fp32 oracle: {N} frames x 2 profiles in {time.time() - t0:.1f} s", flush=True)
    for v in ("0", "1", "2"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, ISB_STEM_SILU=v), capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)


def child():
    import torch
    from isbfsar_amd.hpe_engine import HpeEngine
    from oracle import hpe_oracle as ho
    ref = np.load(REFS)
    n = int(ref["n"])
    W = np.load(os.path.join(ROOT, "isbfsar_amd", "assets", "32_to_122.npy"))
    fr, bb = synth.frames(n, seed=0), synth.bboxes(n, seed=0)
    crops = np.stack([ho.warp(fr[b], ho.crop_params(bb[b], K())[2][0]) for b in range(n)])
    e = HpeEngine(device=0, max_batch=max(n, 256))
    e.set_joint_map(W, None)
    res = {"ISB_STEM_SILU": os.environ.get("ISB_STEM_SILU")}
    for prof, st in states().items():
        e.load_weights(st)
        joints, valid = e.forward(fr, bb)
        _, lg = e.backbone(crops)
        ok = [j for j in range(n) if ref[prof + "_ok"][j] and valid[j]]
        d = np.sort(np.abs(joints[ok].astype(np.float64) - ref[prof + "_pose"][ok]).reshape(len(ok), -1).max(axis=1))
        d3 = np.abs(ho.decode(lg)[1] - ref[prof + "_p3"]).reshape(n, -1).max(axis=1)
        res[prof] = {"abs_p50": float(np.median(d)), "abs_p90": float(d[int(0.9 * len(d))]), "abs_max": float(d[-1]), "abs_mean": float(d.mean()),
                     "dec3d_p50": float(np.median(d3)), "dec3d_max": float(d3.max()), "frames": len(ok)}
    # the stem's share of a 256-frame pass: time 20 forward passes
    f = torch.from_numpy(synth.frames(256, seed=1)).cuda()
    b = torch.from_numpy(synth.bboxes(256, seed=1)).cuda()
    e.set_lanes(1)
    for _ in range(3):
        e.forward(f, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        e.forward(f, b)
    torch.cuda.synchronize()
    res["ms_per_256_frames_one_lane"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    child() if "--child" in sys.argv else parent()
