import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from isbfsar_amd.engine import ArEngine
from isbfsar_amd import synth, weights
gd = "tests/golden"
def sm(x):
    e = np.exp(x - x.max(-1, keepdims=True)); return e / e.sum(-1, keepdims=True)
for name in ("ar_ref_16_30_5.npz", "ar_bl_30_122_60.npz", "ar_bl_30_122_120.npz", "ar_sharp_16_30_5.npz", "ar_sharp_30_122_60.npz"):
    g = np.load(os.path.join(gd, name))
    L, J, way, B, seed = (int(g[k]) for k in ("L", "J", "way", "B", "seed"))
    kw = dict(disc_gain=float(g["disc_gain"]), norm_gain=float(g["norm_gain"])) if "sharp" in name else {}
    state = weights.make_ar_state(L, J, seed=seed, **kw)
    ss = synth.skeleton_windows(way, L, J, seed=seed + 100); q = synth.skeleton_windows(B, L, J, seed=seed + 200)
    out = []
    for prec in ("bf16", "f16", "bf16x3"):
        e = ArEngine(L, J, way, device=0, precision=prec); e.load_weights(state); e.set_support(poses=ss)
        lg, it, _ = e.infer(q)
        out.append(f"{prec}: dlogit {np.abs(lg - g['logits']).max():.2e} dprob {np.abs(sm(lg) - sm(g['logits'])).max():.2e} dis_true {np.abs(it - g['is_true'][:, 0]).max():.2e}")
        e.close()
    print(name, f"|logit|max {np.abs(g['logits']).max():.1f}", " | ".join(out), flush=True)
