"""The vector-ALU floor of the pose backbone beside its matrix floor (CPU only).
EfficientNetV2-L is elementwise-heavy: every expanded element passes two SiLUs (two quarter-rate transcendentals each) and nine
depthwise taps, every gated fragment element an unpack / multiply / convert. A CDNA SIMD retires one 64-lane vector instruction per
4 cycles (16 for v_exp_f32 / v_rcp_f32), so the elementwise work has a floor of its own, which no tiling of the GEMMs removes.
usage: python tools/valu_floor.py [frames] [GHz]"""
import sys

sys.path.insert(0, ".")
from isbfsar_amd import effnetv2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
clk = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 2.1e9
SILU = 4 * 4.5 + 2 * 16      # + bias, x scale, + 1, x, half a packed convert; v_exp_f32 + v_rcp_f32
TAPS = 9 * 4                  # v_dot2 per tap and channel
GATE = 2.5 * 4                # unpack / packed multiply / convert per gated element (each fragment is gated by the two wave columns reading it)
EPI = 2.5 * 4                 # bias + residual unpack / add + convert of a projection output
silu = taps = gate = epi = macs = 0
per_stage = {}
for b in effnetv2.blocks():
    o, i = b.out_hw ** 2, b.in_hw ** 2
    if b.kind == "fused":
        if b.cexp == b.cin:
            s, t, g, e, m = o * b.cout, 0, 0, o * b.cout, o * 9 * b.cin * b.cout
        else:
            s, t, g, e, m = o * b.cexp, 0, 0, o * b.cout, o * 9 * b.cin * b.cexp + o * b.cexp * b.cout
    else:
        s, t, g, e, m = i * b.cexp + o * b.cexp, o * b.cexp, 2 * o * b.cexp, o * b.cout, i * b.cin * b.cexp + o * b.cexp * b.cout
    silu += s; taps += t; gate += g; epi += e; macs += m
    key = f"{b.in_hw}x{b.in_hw} {b.kind}"
    c = (s * SILU + t * TAPS + g * GATE + e * EPI) / 64.0
    ps = per_stage.setdefault(key, [0.0, 0])
    ps[0] += c; ps[1] += m
silu += 128 * 128 * 32 + 64 * 1280
cyc = (silu * SILU + taps * TAPS + gate * GATE + epi * EPI) / 64.0
t_valu = cyc * B / 1024 / clk
t_mfma = 2 * macs * B / 2.5e15
print(f"per frame: {silu / 1e6:.1f} M SiLU evaluations, {taps / 1e6:.1f} M depthwise outputs (x 9 taps), {gate / 1e6:.1f} M gated fragment elements, "
      f"{epi / 1e6:.1f} M projection outputs; {macs / 1e9:.2f} GMAC in the convolutions")
print(f"{B} frames: vector-ALU floor {t_valu * 1e3:.2f} ms (1024 SIMDs, one 64-lane instruction per 4 cycles, {clk / 1e9:.2f} GHz); "
      f"matrix floor {t_mfma * 1e3:.2f} ms (2.5 PFLOP/s dense); ratio {t_valu / t_mfma:.2f}")
for k, (c, m) in per_stage.items():
    print(f"  {k:14s} vector-ALU floor {c * B / 1024 / clk * 1e3:6.2f} ms   matrix floor {2 * m * B / 2.5e15 * 1e3:6.2f} ms")
