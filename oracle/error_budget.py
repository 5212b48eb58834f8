"""ORACLE-SIDE STUDY (test infrastructure, NOT product code): per-stage error budget of the pose stage.

Which bf16 rounding point of the EfficientNetV2-L forward pass sets the distance between the bf16 path and the
fp32 path on what hpe.py:171 returns (the absolute pose, misc.py:141-204) and on what the AR stage consumes (the
root-centred pose, main.py:103)? The fp32 oracle is run once; then the oracle is run with bf16 storage switched on
at ONE rounding point / stage at a time (effnetv2_oracle.EffNetV2LOracle(rounding=...)), and with the candidate
mixed layouts (fp16 in the last stages, f32 or hi+lo residual stream in the MBConv stages, ...). CPU only, torch fp32 convolutions.

    python -m oracle.error_budget [--frames 4] [--profile default|signal|both] [--quick]
    python -m oracle.error_budget --silu [--frames 8]       # cheaper SiLU arithmetic under the product's storage layout (round 5)

Writes a table to stdout (kept in DESIGN.md section 4 and profiles/r03_pose_error_budget.txt).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from isbfsar_amd import effnetv2, synth            # noqa: E402  (weights generator + synthetic inputs only)
from oracle import hpe_oracle as ho                 # noqa: E402
from oracle.effnetv2_oracle import ROUND_POINTS, EffNetV2LOracle   # noqa: E402


def _inputs(n, seed=0):
    K = ho.intrinsics_matrix(384.025146484375, 384.025146484375, 319.09661865234375, 237.75723266601562)
    fr, bb = synth.frames(n, seed=seed), synth.bboxes(n, seed=seed)
    crops, nks, rs = [], [], []
    for i in range(n):
        nk, r, H = ho.crop_params(bb[i], K)
        crops.append(ho.warp(fr[i], H[0]))
        nks.append(nk); rs.append(r)
    return np.stack(crops), nks, rs


def _poses(net, crops, nks, rs, W, idx):
    lg = net.head(net.backbone(crops))
    out = []
    for i in range(len(crops)):
        p2, p3 = ho.decode(lg[i:i + 1])
        pose = ho.postprocess(lg[i:i + 1], nks[i], rs[i], W, idx)
        out.append((p2, p3, pose))
    return out


def _errs(a, b):
    e = {"pred3d": 0.0, "pred2d_px": 0.0, "rc": 0.0, "abs": 0.0}
    for (p2a, p3a, pa), (p2b, p3b, pb) in zip(a, b):
        e["pred3d"] = max(e["pred3d"], float(np.abs(p3a - p3b).max()))
        e["pred2d_px"] = max(e["pred2d_px"], float(np.abs(p2a - p2b).max()))
        if pa is not None and pb is not None:
            e["rc"] = max(e["rc"], float(np.abs((pa - pa[0]) - (pb - pb[0])).max()))
            e["abs"] = max(e["abs"], float(np.abs(pa - pb).max()))
    return e


def policies(quick=False):
    """name -> rounding callable. 'only X' = bf16 at X, f32 everywhere else; 'all but X' = the converse."""
    P = {}
    P["bf16 everywhere (precision 1 = round-2 layout)"] = lambda s, p: "bf16"
    P["bf16, fp16 weights + activations in stages 5-6 + head conv (PRODUCT)"] = lambda s, p: "f16" if s >= 5 else "bf16"
    P["bf16, fp16 activations (bf16 weights) in stages 5-6"] = lambda s, p: "f16" if (s >= 5 and p != "w") else "bf16"
    P["bf16, fp16 residual stream only in stages 5-6"] = lambda s, p: "f16" if (s >= 5 and p == "out") else "bf16"
    P["bf16, fp16 weights + activations in stages 3-6 + head conv"] = lambda s, p: "f16" if s >= 3 else "bf16"
    P["fp16 everywhere (the reference's TensorRT precision)"] = lambda s, p: "f16"
    P["bf16, residual stream hi+lo in stages 5-6 (+15 % pose time)"] = lambda s, p: "bf16x2" if (p == "out" and s >= 5) else "bf16"
    P["only weights bf16"] = lambda s, p: "bf16" if p == "w" else "f32"
    P["only activations bf16"] = lambda s, p: "f32" if p == "w" else "bf16"
    for pt in ROUND_POINTS[1:]:
        P[f"only '{pt}' tensors bf16 (all stages)"] = (lambda pt: lambda s, p: "bf16" if p == pt else "f32")(pt)
    if not quick:
        for st in range(-1, 8):
            P[f"only stage {st} bf16 (weights + activations)"] = (lambda st: lambda s, p: "bf16" if s == st else "f32")(st)
    # candidate layouts
    P["bf16, residual stream hi+lo in stages 4-6"] = lambda s, p: "bf16x2" if (p == "out" and s >= 4) else "bf16"
    P["bf16, f32 stream + bf16 conv copy in stages 3-6"] = lambda s, p: "f32+bf16copy" if (p == "out" and s >= 3) else "bf16"
    P["bf16, all activations hi+lo in stages 5-6"] = lambda s, p: "bf16x2" if (p != "w" and s >= 5) else "bf16"
    P["bf16, residual stream f32 in stages 3-6"] = lambda s, p: "f32" if (p == "out" and s >= 3) else "bf16"
    P["bf16, residual stream hi+lo in stages 3-6"] = lambda s, p: "bf16x2" if (p == "out" and s >= 3) else "bf16"
    P["bf16, residual stream f32 in all stages"] = lambda s, p: "f32" if p == "out" else "bf16"
    P["bf16, residual stream + gate f32 in stages 3-6"] = lambda s, p: "f32" if (p in ("out", "gate") and s >= 3) else "bf16"
    P["bf16, all activations f32 in stages 3-6"] = lambda s, p: "f32" if (p != "w" and s >= 3) else "bf16"
    P["bf16, stages 5-6 + head conv entirely f32"] = lambda s, p: "f32" if s >= 5 else "bf16"
    P["bf16, stage 6 + head conv entirely f32"] = lambda s, p: "f32" if s >= 6 else "bf16"
    return P


def silu_forms():
    """Cheaper SiLUs priced BEFORE anyone builds them: what does the pose lose if the activation's two
    transcendentals and three plain operations (x * rcp(1 + exp2(-x log2 e)), f32: ~23 issue cycles per element and wave, 37 % of the
    fused 16x16 front's time) run in a cheaper arithmetic? Every form is the same expression; what changes is where its intermediate
    results are rounded. name -> callable(x: f32 tensor) -> f32 tensor (the value that is then rounded to the 16-bit storage type)."""
    import torch
    h = lambda t: t.clamp(-65504.0, 65504.0).half().float()       # one fp16 rounding (saturating like the device)
    b = lambda t: t.bfloat16().float()
    L2E = 1.4426950408889634

    def exact(x):
        return x * torch.sigmoid(x)

    def f32_fast(x):                  # the product: v_exp_f32 / v_rcp_f32 (each ~1 ulp), f32 multiplies
        return x * (1.0 / (1.0 + torch.exp2(-L2E * x)))

    def f16_all(x):                   # everything in fp16: v_cvt, v_pk_mul_f16, v_exp_f16, v_pk_add_f16, v_rcp_f16, v_pk_mul_f16
        xh = h(x)
        e = h(torch.exp2(h(-L2E * xh)))
        return h(xh * h(1.0 / h(1.0 + e)))

    def f16_transc(x):                # f32 multiplies and add, only the two transcendentals in fp16 (v_exp_f16 / v_rcp_f16)
        e = h(torch.exp2(h(-L2E * x)))
        return x * h(1.0 / h(1.0 + e))

    def f16_rcp(x):                   # v_exp_f32, the reciprocal in fp16
        return x * h(1.0 / h(1.0 + torch.exp2(-L2E * x)))

    def bf16_gate(x):                 # sigmoid rounded to bf16 before the multiply (an 8-bit gate)
        return x * b(1.0 / (1.0 + torch.exp2(-L2E * x)))

    return {"exact x * sigmoid(x) in f32 (the definition)": exact,
            "f32 exp2 / rcp (PRODUCT: silu_fast)": f32_fast,
            "exp2 and rcp in fp16, f32 multiplies": f16_transc,
            "rcp in fp16 only": f16_rcp,
            "everything in fp16 (packed fp16 multiplies, fp16 transcendentals)": f16_all,
            "sigmoid rounded to bf16": bf16_gate}


def silu_study(a, crops, nks, rs, W, idx):
    """The product's storage layout (fp16 everywhere) with each SiLU form, against the fp32 definition, on both weight profiles."""
    from oracle import effnetv2_oracle as eo
    pol = lambda s, p: "f16"
    res = {}
    for prof in (["default", "signal"] if a.profile == "both" else [a.profile]):
        # the "signal" profile as tests/test_hpe_gpu.py and bench.py judge it: head gain 0.5 (peaked but not one-hot heat-maps)
        state = effnetv2.make_state(0, profile=prof) if prof == "default" else effnetv2.make_state(0, prof, head_gain=0.5)
        base = _poses(EffNetV2LOracle(state, "f32"), crops, nks, rs, W, idx)
        print(f"# profile {prof}: {a.frames} frames; storage fp16 in every stage; columns: max-abs against the fp32 definition", flush=True)
        print(f"{'SiLU form':72s} {'pred3d':>9s} {'pred2d px':>9s} {'root-c.':>9s} {'absolute':>9s}")
        res[prof] = {}
        keep = eo._silu
        try:
            for name, fn in silu_forms().items():
                eo._silu = fn
                e = _errs(_poses(EffNetV2LOracle(state, "f32", rounding=pol), crops, nks, rs, W, idx), base)
                res[prof][name] = e
                print(f"{name:72s} {e['pred3d']:9.2e} {e['pred2d_px']:9.2e} {e['rc']:9.2e} {e['abs']:9.2e}", flush=True)
        finally:
            eo._silu = keep
    return res


def f16_points_study(a):
    """Round 6: the product's layout IS fp16 everywhere; which of its rounding points carries the distance to the
    fp32 definition on the hard ("signal") profile, as a DISTRIBUTION over frames (all 122 joints: what bench.py judges)? fp16 storage
    switched on at one point / in one group of stages at a time, exact SiLU and the product's SiLU."""
    from oracle import effnetv2_oracle as eo
    assets = os.path.join(ROOT, "isbfsar_amd", "assets")
    W = np.load(os.path.join(assets, "32_to_122.npy"))
    crops, nks, rs = _inputs(a.frames)
    forms = silu_forms()
    exact, fast = forms["exact x * sigmoid(x) in f32 (the definition)"], forms["f32 exp2 / rcp (PRODUCT: silu_fast)"]
    P = {"fp16 everywhere, exact SiLU": (lambda s, p: "f16", exact),
         "fp16 everywhere, product SiLU (v_exp / v_rcp)": (lambda s, p: "f16", fast),
         "f32 storage, product SiLU": (lambda s, p: "f32", fast)}
    for pt in ROUND_POINTS:
        P[f"only '{pt}' fp16"] = ((lambda pt: lambda s, p: "f16" if p == pt else "f32")(pt), exact)
    for lo, hi, nm in ((-1, 2, "stem + stages 0-2"), (3, 4, "stages 3-4 (16x16 maps)"), (5, 7, "stages 5-6 + head conv (8x8 maps)")):
        P[f"only {nm} fp16"] = ((lambda lo, hi: lambda s, p: "f16" if lo <= s <= hi else "f32")(lo, hi), exact)
    P["fp16 everywhere but the residual stream (f32 'out')"] = (lambda s, p: "f32" if p == "out" else "f16", exact)
    P["fp16 everywhere but the weights"] = (lambda s, p: "f32" if p == "w" else "f16", exact)
    for prof in (["default", "signal"] if a.profile == "both" else [a.profile]):
        state = effnetv2.make_state(0, profile=prof) if prof == "default" else effnetv2.make_state(0, prof, head_gain=0.5)
        base = _poses(EffNetV2LOracle(state, "f32"), crops, nks, rs, W, None)
        print(f"# profile {prof}: {a.frames} frames, 122 joints; |absolute pose - fp32 definition| per frame: p50 / p90 / max; decoded 3D max", flush=True)
        keep = eo._silu
        try:
            for name, (pol, fn) in P.items():
                if a.only and a.only not in name:
                    continue
                eo._silu = fn
                got = _poses(EffNetV2LOracle(state, "f32", rounding=pol), crops, nks, rs, W, None)
                d = np.sort([float(np.abs(g[2] - b[2]).max()) for g, b in zip(got, base) if g[2] is not None and b[2] is not None])
                d3 = max(float(np.abs(g[1] - b[1]).max()) for g, b in zip(got, base))
                print(f"{name:60s} {np.median(d):9.2e} {d[int(0.9 * (len(d) - 1))]:9.2e} {d[-1]:9.2e}   {d3:9.2e}", flush=True)
        finally:
            eo._silu = keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="--f16points: only the layouts whose name holds this substring")
    ap.add_argument("--f16points", action="store_true", help="round 6: the fp16 layout's distance to fp32 by rounding point, per-frame distribution")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--profile", default="both")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--silu", action="store_true", help="price cheaper SiLU arithmetic under the product's fp16 storage layout")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    if a.f16points:
        f16_points_study(a)
        return
    if a.silu:
        assets = os.path.join(ROOT, "isbfsar_amd", "assets")
        W = np.load(os.path.join(assets, "32_to_122.npy"))
        idx = json.load(open(os.path.join(assets, "skeleton_types.json")))["smpl+head_30"]["indices"]
        crops, nks, rs = _inputs(a.frames)
        res = silu_study(a, crops, nks, rs, W, idx)
        if a.json:
            with open(a.json, "w") as f:
                json.dump(res, f, indent=1)
        return
    assets = os.path.join(ROOT, "isbfsar_amd", "assets")
    W = np.load(os.path.join(assets, "32_to_122.npy"))
    idx = json.load(open(os.path.join(assets, "skeleton_types.json")))["smpl+head_30"]["indices"]
    crops, nks, rs = _inputs(a.frames)
    res = {}
    for prof in (["default", "signal"] if a.profile == "both" else [a.profile]):
        state = effnetv2.make_state(0, profile=prof)
        t0 = time.time()
        base = _poses(EffNetV2LOracle(state, "f32"), crops, nks, rs, W, idx)
        print(f"# profile {prof}: {a.frames} frames, fp32 pass {time.time() - t0:.1f} s", flush=True)
        print(f"{'rounding layout':64s} {'pred3d':>9s} {'pred2d px':>9s} {'root-c.':>9s} {'absolute':>9s}")
        res[prof] = {}
        for name, pol in policies(a.quick).items():
            e = _errs(_poses(EffNetV2LOracle(state, "f32", rounding=pol), crops, nks, rs, W, idx), base)
            res[prof][name] = e
            print(f"{name:64s} {e['pred3d']:9.2e} {e['pred2d_px']:9.2e} {e['rc']:9.2e} {e['abs']:9.2e}", flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
