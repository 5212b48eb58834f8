"""Per-layer-group time of the convolution launches of the LAST backbone pass in a rocprofv3 kernel trace (one lane).
usage: python tools/layer_breakdown.py <kernel_trace.csv> [B]
The expected launch sequence follows csrc/hpe_api.cpp::run_backbone: Fused-MBConv blocks with <= 256 expanded channels
are ONE launch (fused_mb), the others expand + project; MBConv blocks are expand + project (depthwise / SE are not
convolution-family launches), except that the stride-1 MBConv blocks with 384 inputs on 8 x 8 maps (mbfront8) and with 192 / 224 inputs on
16 x 16 maps (mbfront16, round 5) run expand + depthwise + pool as one launch.
Each row also carries the layer's two floors: MFMA (its FLOPs at the 2.5 PFLOP/s dense bf16 peak) and HBM (its algorithmic bytes --
input + output [+ residual] activations and the weights, 2 bytes per element -- at 8 TB/s), and `eff` = the larger floor / the
measured time. The last line is the pass at the speed of light of this launch structure (sum of the larger floors)."""
import collections
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isbfsar_amd import effnetv2 as E

FAMILY = ("conv_igemm", "gemm1x1", "conv3x3_dma", "conv3x3_c32_rows", "fused_mb", "mbfront8", "mbfront16", "splitk_reduce")
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in FAMILY)]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
PEAK_F, PEAK_B = 2.5e15, 8.0e12
seq = []                      # (label, flops, algorithmic bytes)
for b in E.blocks():
    o = b.out_hw * b.out_hw
    if b.kind == "fused":
        if b.cexp == b.cin:
            seq.append((f"f3x3 {b.cin}->{b.cout} @{b.out_hw}", 2.0 * B * o * 9 * b.cin * b.cout,
                        2.0 * (B * b.in_hw ** 2 * b.cin + B * o * b.cout * (2 if b.residual else 1) + 9 * b.cin * b.cout)))
        elif b.cexp <= 256 and b.cout <= 128:
            seq.append((f"fusedMB {b.cin}->{b.cexp}->{b.cout} @{b.out_hw} s{b.stride}", 2.0 * B * o * (9 * b.cin * b.cexp + b.cexp * b.cout),
                        2.0 * (B * b.in_hw ** 2 * b.cin + B * o * b.cout * (2 if b.residual else 1) + 9 * b.cin * b.cexp + b.cexp * b.cout)))
        else:
            seq.append((f"exp3x3 {b.cin}->{b.cexp} @{b.out_hw}", 2.0 * B * o * 9 * b.cin * b.cexp,
                        2.0 * (B * b.in_hw ** 2 * b.cin + B * o * b.cexp + 9 * b.cin * b.cexp)))
            seq.append((f"proj {b.cexp}->{b.cout} @{b.out_hw}", 2.0 * B * o * b.cexp * b.cout,
                        2.0 * (B * o * b.cexp + B * o * b.cout * (2 if b.residual else 1) + b.cexp * b.cout)))
    else:
        # stride-1 blocks with 384 inputs on 8 x 8 maps: expand + depthwise + pool are ONE launch (mbfront8_kernel) -- its time holds the
        # depthwise work, its floors here are the expand GEMM's (the expanded tensor is written once, as the depthwise output)
        front = b.stride == 1 and B >= 32 and ((b.in_hw == 8 and b.cin == 384) or (b.in_hw == 16 and b.cin in (192, 224)))       # mbfront8 / mbfront16
        seq.append((f"{'front(exp+dw)' if front else 'exp1x1'} {b.cin}->{b.cexp} @{b.in_hw}", 2.0 * B * b.in_hw * b.in_hw * b.cin * b.cexp,
                    2.0 * (B * b.in_hw ** 2 * (b.cin + b.cexp) + b.cin * b.cexp)))
        seq.append((f"proj(SE) {b.cexp}->{b.cout} @{b.out_hw}", 2.0 * B * o * b.cexp * b.cout,
                    2.0 * (B * o * b.cexp + B * o * b.cout * (2 if b.residual else 1) + b.cexp * b.cout) + 4.0 * B * b.cexp))
seq.append(("head 640->1280 @8", 2.0 * B * 64 * 640 * 1280, 2.0 * (B * 64 * 640 + 640 * 1280) + 4.0 * B * 64 * 1280))
n = len(seq)
last = rows[-n:]
agg = collections.OrderedDict()
tot = 0.0
sol = 0.0
for (label, fl, by), r in zip(seq, last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(label, [0.0, 0.0, 0, r["Kernel_Name"].replace("void isb::", "").replace("(anonymous namespace)::", "")[:36], 0.0, 0.0])
    a[0] += d; a[1] += fl; a[2] += 1; a[4] += by
    a[5] += max(fl / PEAK_F, by / PEAK_B) * 1e3
    tot += d
    sol += max(fl / PEAK_F, by / PEAK_B) * 1e3
print(f"convolution-family launches in the trace: {len(rows)}; per pass {n}; last pass {tot:.3f} ms "
      f"({sum(f for _, f, _ in seq) / tot / 1e9:.0f} TFLOP/s over the pass), B={B}")
for k, a in agg.items():
    mf, hb = a[1] / PEAK_F * 1e3, a[4] / PEAK_B * 1e3
    print(f"{k:34s} n={a[2]:2d} ms={a[0]:6.3f} ({100 * a[0] / tot:4.1f}%) TFLOP/s={a[1] / a[0] / 1e9:6.0f}  floors mfma {mf:5.3f} hbm {hb:5.3f} ms"
          f" ({'hbm' if hb > mf else 'mfma'}-bound)  eff={a[5] / a[0]:4.2f}  {a[3]}")
print(f"the pass at the speed of light of this launch structure (per layer the larger of its MFMA and HBM floors): {sol:.3f} ms = "
      f"{sol / tot:.2f} of the measured {tot:.3f} ms")
