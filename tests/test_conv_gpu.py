"""GPU unit tests of the conv_igemm kernel family (every tile variant) against a torch-CPU conv on
the same bf16-rounded operands: 3x3 / 1x1, stride 1 / 2 (TF SAME), SiLU, residual, SE gate."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from isbfsar_amd.hpe_engine import bf16_to_f32, conv_debug, f32_to_bf16

pytestmark = pytest.mark.gpu


def _ref(x, w, scale, shift, k, stride, act, res, gate):
    xb = torch.from_numpy(bf16_to_f32(f32_to_bf16(x)))                       # [B,H,W,Cin]
    if gate is not None:
        xb = (xb * torch.from_numpy(gate)[:, None, None, :]).bfloat16().float()
    wf = (torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1, 1)).bfloat16().float()
    xi = xb.permute(0, 3, 1, 2)
    wi = wf.permute(0, 3, 1, 2)
    if k == 3 and stride == 2:
        y = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wi, stride=2)
    elif k == 3:
        y = F.conv2d(xi, wi, padding=1)
    else:
        y = F.conv2d(xi, wi, stride=stride)
    y = y + torch.from_numpy(shift).view(1, -1, 1, 1)
    if act:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + torch.from_numpy(bf16_to_f32(f32_to_bf16(res)))
    return y.bfloat16().float().numpy()


CASES = [
    # B, H, Cin, Cout, k, stride, act, res, gate
    (2, 16, 64, 128, 3, 1, 1, False, False),
    (2, 16, 32, 128, 3, 2, 1, False, False),
    (3, 8, 96, 96, 1, 1, 0, True, True),
    (2, 16, 64, 224, 1, 1, 0, True, True),
    (2, 16, 32, 32, 3, 1, 1, True, False),
    (1, 16, 128, 64, 1, 1, 0, False, False),
    (2, 8, 192, 256, 1, 1, 1, False, False),
    (5, 8, 256, 384, 1, 1, 0, True, True),
    (3, 16, 96, 64, 1, 1, 0, True, False),        # un-gated projection with residual, 3 k-steps, ragged M
    (4, 32, 160, 224, 1, 1, 1, False, False),     # 5 k-steps (odd), 224 = 7 x 32 output channels
    (3, 16, 96, 160, 3, 1, 1, False, False),      # 3x3, 3 k-steps per tap, ragged M (768 rows), Cout tile overhang
    (2, 32, 64, 64, 3, 2, 0, True, False),        # 3x3 stride 2 (TF SAME), residual, no activation
    (2, 16, 192, 224, 1, 1, 0, True, True),       # gated projections with Cin % 96 == 0 (the three-buffer kernels): 6 k-steps
    (3, 8, 288, 384, 1, 1, 0, True, True),        # 9 k-steps, ragged M
]
# The tile variants of the product build (what launch_conv_igemm selects for the pose backbone, the detector and the ResNet
# trunk; the ~130 measured-and-rejected shapes exist only in ISB_BUILD_PROBES=1 builds, tools/conv_sweep.py runs them there)
GENERAL = [1, 3, 5, 75]                         # register-staged fallback (takes a gate)
DMA = [14, 54, 55, 57, 59, 64]                  # general LDS-DMA kernel (no gate)
G1 = [131, 132, 135, 138, 150]                  # lean 1x1 GEMM kernels
G1G = [141, 142, 143, 144, 146, 147]            # SE-gated 1x1 GEMMs
C3 = [161, 162, 163, 164, 165, 168, 169]             # lean 3x3 kernels (buffer-addressed A operand)


def _variant_takes(variant, case):
    B, H, Cin, Cout, k, stride, act, use_res, use_gate = case
    if variant in DMA:
        return not use_gate
    if variant in C3:
        return not use_gate and k == 3
    if variant in G1:
        return not use_gate and k == 1 and stride == 1
    if variant in G1G:
        ohw, bm = H * H, 64 if variant in (146, 147) else 128
        return use_gate and (ohw % bm == 0 or bm % ohw == 0)
    return True


ALL = [0] + GENERAL + DMA + G1 + G1G + C3


@pytest.mark.parametrize("case,variant", [(c, v) for c in CASES for v in ALL if _variant_takes(v, c)])
def test_conv_variants(case, variant):
    B, H, Cin, Cout, k, stride, act, use_res, use_gate = case
    rng = np.random.default_rng(hash((case, 7)) % (2 ** 31))
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    OH = H // stride
    res = rng.normal(0, 1, (B, OH, OH, Cout)).astype(np.float32) if use_res else None
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32) if use_gate else None
    out, ms = conv_debug(f32_to_bf16(x), w, scale, shift, k, stride, act,
                         None if res is None else f32_to_bf16(res), gate, variant=variant)
    got = bf16_to_f32(out)
    ref = _ref(x, w, scale, shift, k, stride, act, res, gate)
    # both sides round to bf16 once; accumulation order differs -> at most 1 bf16 ulp apart
    tol = 2.0 ** -7 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())
    assert ms >= 0


@pytest.mark.parametrize("B,H,act,use_res", [(1, 128, 1, True), (3, 128, 1, False), (2, 6, 0, True)])
def test_conv_halo_c32(B, H, act, use_res):
    """3x3 32 -> 32 on 128-wide images from an LDS halo tile (variant 171): against the torch reference and, bit for
    bit, against the implicit-GEMM kernel (variant 163) -- same (tap, channel) summation order."""
    rng = np.random.default_rng(1000 * B + H)
    x = rng.normal(0, 1, (B, H, 128, 32)).astype(np.float32)
    w = (rng.normal(0, 1, (32, 3, 3, 32)) / np.sqrt(288)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, 32).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, 32).astype(np.float32)
    res = rng.normal(0, 1, (B, H, 128, 32)).astype(np.float32) if use_res else None
    r16 = None if res is None else f32_to_bf16(res)
    out, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 3, 1, act, r16, None, variant=171)
    old, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 3, 1, act, r16, None, variant=163)
    assert np.array_equal(out, old)
    xb = torch.from_numpy(bf16_to_f32(f32_to_bf16(x))).permute(0, 3, 1, 2)
    wf = (torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1, 1)).bfloat16().float().permute(0, 3, 1, 2)
    y = F.conv2d(xb, wf, padding=1) + torch.from_numpy(shift).view(1, -1, 1, 1)
    if act:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + torch.from_numpy(bf16_to_f32(r16))
    ref = y.bfloat16().float().numpy()
    got = bf16_to_f32(out)
    tol = 2.0 ** -7 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 32, 32, 64), (3, 16, 64, 128), (2, 16, 128, 256), (5, 14, 128, 128), (1, 8, 512, 1024)])
def test_conv3x3_stride2_symmetric_padding(B, H, Cin, Cout):
    """The detector's / ResNet trunk's down-sampling layers (PyTorch: stride 2, pad 1 on every side) on the lean 3x3 kernel -- its
    taps come from a shifted buffer base and a per-lane validity mask, so the padding rule is only the window origin: against
    torch-CPU on the same rounded operands, and the same bits as the general implicit-GEMM kernel (variant 55)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(B + H + Cin + Cout)
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    out, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 3, 2, 1, torch_pad=True)
    gen, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 3, 2, 1, variant=55, torch_pad=True)
    xb = torch.from_numpy(bf16_to_f32(f32_to_bf16(x))).permute(0, 3, 1, 2)
    wf = (torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1, 1)).bfloat16().float().permute(0, 3, 1, 2)
    y = F.conv2d(xb, wf, stride=2, padding=1) + torch.from_numpy(shift).view(1, -1, 1, 1)
    ref = (y * torch.sigmoid(y)).permute(0, 2, 3, 1).bfloat16().float().numpy()
    got = bf16_to_f32(out)
    assert got.shape == ref.shape
    tol = 2.0 ** -7 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())
    assert np.array_equal(out, gen)


@pytest.mark.parametrize("B,Cout,act", [(2, 384, 1), (3, 192, 0)])
def test_conv3x3_halo_c96(B, Cout, act):
    """3x3 96 -> Cout on 32 x 32 maps with the A operand read from an LDS halo tile (variant 167): bit for bit the
    implicit-GEMM kernel (variant 161), and the torch reference."""
    rng = np.random.default_rng(B * 100 + Cout)
    x = rng.normal(0, 1, (B, 32, 32, 96)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 3, 3, 96)) / np.sqrt(864)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    out, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 3, 1, act, None, None, variant=167)
    old, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 3, 1, act, None, None, variant=161)
    assert np.array_equal(out, old)
    ref = _ref(x, w, scale, shift, 3, 1, act, None, None)
    got = bf16_to_f32(out)
    tol = 2.0 ** -7 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())


SPLITK_CASES = [
    # B, H, Cin, Cout, act, res, gate, splits, tile variant
    (1, 8, 2304, 384, 0, True, True, 12, 147),     # stage-6 projection of one frame
    (1, 8, 3840, 640, 0, True, True, 16, 147),     # stage-7 projection: 120 k-tiles
    (1, 16, 1344, 224, 0, True, True, 7, 147),     # stage-5 projection, 42 k-tiles in 7 splits
    (2, 8, 416, 192, 0, False, True, 5, 147),      # 13 k-tiles in 5 splits -> ceil 3 per split, 5 splits, last one short
    (3, 8, 320, 128, 1, True, False, 4, 138),      # un-gated, SiLU in the reduction, ragged M (192 rows)
    (1, 8, 224, 64, 0, False, False, 16, 138),     # more splits asked for than k-tiles (7): clamped, none empty
    (2, 8, 96, 96, 0, True, False, 3, 132),        # one k-tile per split on a 128 x 128 tile with overhang
]


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_conv_split_k(case):
    """split-K GEMM + reduction (single-frame projections) against the torch reference and the one-launch kernel."""
    B, H, Cin, Cout, act, use_res, use_gate, splits, variant = case
    rng = np.random.default_rng(hash((case, 11)) % (2 ** 31))
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    res = rng.normal(0, 1, (B, H, H, Cout)).astype(np.float32) if use_res else None
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32) if use_gate else None
    r16 = None if res is None else f32_to_bf16(res)
    out, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 1, 1, act, r16, gate, variant=1000 * splits + variant)
    one, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 1, 1, act, r16, gate, variant=variant)
    got, single = bf16_to_f32(out), bf16_to_f32(one)
    ref = _ref(x, w, scale, shift, 1, 1, act, res, gate)
    tol = 2.0 ** -7 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())
    assert np.all(np.abs(got - single) <= tol)
    # the reduction adds the partials in split order: repeated launches are bit-identical
    again, _ = conv_debug(f32_to_bf16(x), w, scale, shift, 1, 1, act, r16, gate, variant=1000 * splits + variant)
    assert np.array_equal(out, again)


@pytest.mark.parametrize("B,H,C,stride", [(2, 16, 768, 1), (3, 8, 2304, 1), (2, 16, 1344, 2), (1, 32, 384, 2), (2, 8, 3840, 1)])
def test_depthwise_pool(B, H, C, stride):
    """Depthwise 3x3 + folded BN + SiLU + SE mean (v_dot2c_f32_bf16 on bf16 taps) vs torch-CPU on the same rounded taps."""
    from isbfsar_amd.hpe_engine import dwconv_debug
    rng = np.random.default_rng(B * 100 + C + stride)
    x = rng.normal(0, 1, (B, H, H, C)).astype(np.float32)
    w = (rng.normal(0, 1, (C, 3, 3)) / 3).astype(np.float32)
    sc = rng.uniform(0.8, 1.2, C).astype(np.float32)
    sh = rng.uniform(-0.1, 0.1, C).astype(np.float32)
    out, pooled, _ = dwconv_debug(f32_to_bf16(x), w, sc, sh, stride)
    xb = torch.from_numpy(bf16_to_f32(f32_to_bf16(x))).permute(0, 3, 1, 2)
    wd = (torch.from_numpy(w) * torch.from_numpy(sc)[:, None, None]).bfloat16().float().unsqueeze(1)
    if stride == 2:
        d = F.conv2d(F.pad(xb, (0, 1, 0, 1)), wd, stride=2, groups=C)
    else:
        d = F.conv2d(xb, wd, padding=1, groups=C)
    d = d + torch.from_numpy(sh).view(1, -1, 1, 1)
    ref = (d * torch.sigmoid(d)).bfloat16().float().permute(0, 2, 3, 1).numpy()
    got = bf16_to_f32(out)
    tol = 2.0 ** -7 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())
    assert np.mean(got != ref) < 0.02                       # identical bf16 values except at rounding boundaries
    np.testing.assert_allclose(pooled, got.reshape(B, -1, C).mean(1), rtol=0, atol=1e-5)


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,H,Cin,Cexp,Cout2,stride,use_res", [
    (2, 32, 64, 256, 64, 1, True),      # stage 2 body block
    (2, 64, 64, 256, 64, 1, True),      # stage 2 body block at its real 64 x 64 size: halo-tile A operand
    (1, 64, 64, 256, 64, 1, False),
    (3, 16, 32, 128, 64, 2, False),     # stage 2 first block (stride 2), ragged M (192 rows)
    (1, 32, 64, 256, 96, 2, False),     # stage 3 first block
])
def test_fused_mbconv_block(B, H, Cin, Cexp, Cout2, stride, use_res, f16):
    """One-launch Fused-MBConv block, in bf16 and in fp16 (ConvArgs.f16). With the E tile through LDS (round 2's form, still what
    projections to 96 channels run) it is bit-identical to the two-launch path (lean 3x3 kernel -> 1x1 GEMM kernel). With the
    projection straight from the accumulators (round 5, projections to 64 channels: fused_mb_kernel<.., REGE>) the expanded values
    are the same bits and the projection's sum is associated differently -- (channels 0 - 127) + (128 - 255), the k slots of an MFMA in
    accumulator-row order -- so it agrees with the two-launch path to f32 rounding: the same 16-bit outputs except at rounding
    boundaries. Both within one ulp of the 16-bit type of torch-CPU on the same rounded operands."""
    from isbfsar_amd.hpe_engine import f16_to_f32, f32_to_f16, fused_mb_debug
    rng = np.random.default_rng(Cexp * 10 + Cout2 + stride)
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w1 = (rng.normal(0, 1, (Cexp, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
    s1 = rng.uniform(0.8, 1.2, Cexp).astype(np.float32); b1 = rng.uniform(-0.1, 0.1, Cexp).astype(np.float32)
    w2 = (rng.normal(0, 1, (Cout2, Cexp)) / np.sqrt(Cexp)).astype(np.float32)
    s2 = rng.uniform(0.8, 1.2, Cout2).astype(np.float32); b2 = rng.uniform(-0.1, 0.1, Cout2).astype(np.float32)
    OH = H // stride
    res = rng.normal(0, 1, (B, OH, OH, Cout2)).astype(np.float32) if use_res else None
    cvt, back = (f32_to_f16, f16_to_f32) if f16 else (f32_to_bf16, bf16_to_f32)
    xb, rb = cvt(x), (None if res is None else cvt(res))
    out, _ = fused_mb_debug(xb, w1, s1, b1, w2, s2, b2, rb, stride, f16=f16)
    lds_e, _ = fused_mb_debug(xb, w1, s1, b1, w2, s2, b2, rb, stride, f16=f16, lds_e=True)
    # two-launch path through the same library
    e, _ = conv_debug(xb, w1, s1, b1, 3, stride, 1, None, None, variant=0, f16=f16)
    two, _ = conv_debug(e, w2.reshape(Cout2, 1, 1, Cexp), s2, b2, 1, 1, 0, rb, None, variant=0, f16=f16)
    assert np.array_equal(lds_e, two)
    if Cout2 > 64:
        assert np.array_equal(out, two)                      # (96 projected channels: the LDS form is the only one)
    else:
        fo, ft = back(out), back(two)
        step = (2.0 ** -10 if f16 else 2.0 ** -7) * np.abs(ft) + 2e-6
        assert np.all(np.abs(fo - ft) <= step), float(np.abs(fo - ft).max())
        assert np.mean(out != two) < 5e-3, float(np.mean(out != two))
        again, _ = fused_mb_debug(xb, w1, s1, b1, w2, s2, b2, rb, stride, f16=f16, iters=2)
        assert np.array_equal(out, again)
    # torch-CPU
    got = back(out)
    if f16:
        e_ref = _ref16(x, w1, s1, b1, 3, stride, 1, None, "f16")
        ref = _ref16(e_ref, w2.reshape(Cout2, 1, 1, Cexp), s2, b2, 1, 1, 0, res, "f16")
        ulp = 2.0 ** -10
    else:
        e_ref = _ref(x, w1, s1, b1, 3, stride, 1, None, None)
        ref = _ref(e_ref, w2.reshape(Cout2, 1, 1, Cexp), s2, b2, 1, 1, 0, res, None)
        ulp = 2.0 ** -7
    assert np.mean(np.abs(got - ref) > ulp * np.maximum(1.0, np.abs(ref))) < 0.02
    assert np.abs(got - ref).max() < 4 * ulp * max(1.0, np.abs(ref).max())


def _ref16(x, w, scale, shift, k, stride, act, res, kind):
    """torch-CPU convolution on operands rounded to `kind` ("f16" / "bf16"), one rounding of the result"""
    t = (lambda v: v.half().float()) if kind == "f16" else (lambda v: v.bfloat16().float())
    xb = t(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).permute(0, 3, 1, 2)
    wf = t(torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1, 1)).permute(0, 3, 1, 2)
    if k == 3 and stride == 2:
        y = F.conv2d(F.pad(xb, (0, 1, 0, 1)), wf, stride=2)
    elif k == 3:
        y = F.conv2d(xb, wf, padding=1)
    else:
        y = F.conv2d(xb, wf, stride=stride)
    y = y + torch.from_numpy(shift).view(1, -1, 1, 1)
    if act:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + t(torch.from_numpy(res))
    return t(y).numpy()


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,act,use_res,variant", [
    (2, 16, 128, 32, 32, 1, 1, True, 171),       # stage 0: the row-ring kernel
    (2, 16, 128, 32, 32, 1, 1, True, 163),       # ... and the implicit-GEMM kernel it must equal
    (3, 16, 16, 32, 128, 2, 1, False, 161),      # 3x3 stride 2 (TF SAME)
    (2, 32, 32, 96, 384, 1, 1, False, 167),      # stage 2 body: halo-tile A operand
    (2, 32, 32, 96, 384, 1, 1, False, 161),
    (2, 32, 32, 96, 384, 1, 1, False, 0),
])
def test_conv3x3_f16_operands(B, H, W, Cin, Cout, stride, act, use_res, variant):
    """fp16 forms of the 3x3 kernels (isb_hpe_cfg.precision 0 / 2 runs every stage in fp16): against torch on the same
    fp16-rounded operands, and the row-ring / halo kernels bit for bit against the implicit-GEMM kernel."""
    from isbfsar_amd.hpe_engine import f16_to_f32, f32_to_f16
    rng = np.random.default_rng(B + H + Cin + Cout + stride)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    res = rng.normal(0, 1, (B, H // stride, W // stride, Cout)).astype(np.float32) if use_res else None
    r16 = None if res is None else f32_to_f16(res)
    out, _ = conv_debug(f32_to_f16(x), w, scale, shift, 3, stride, act, r16, None, variant=variant, f16=True)
    got = f16_to_f32(out)
    ref = _ref16(x, w, scale, shift, 3, stride, act, res, "f16")
    assert np.isfinite(got).all()
    tol = 2.0 ** -10 * np.maximum(0.25, np.abs(ref)) + 1e-4
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())
    if variant in (171, 167):
        base, _ = conv_debug(f32_to_f16(x), w, scale, shift, 3, stride, act, r16, None, variant=163 if variant == 171 else 161, f16=True)
        assert np.array_equal(out, base)


@pytest.mark.parametrize("B,H,Cin,Cout,act", [(2, 16, 96, 384, 1), (3, 16, 192, 768, 1), (5, 8, 224, 1344, 1), (1, 8, 224, 192, 0),
                                              (40, 16, 224, 1344, 1), (33, 16, 192, 1152, 1), (7, 32, 96, 384, 1),
                                              (37, 8, 384, 2304, 1), (300, 8, 384, 2304, 1), (130, 16, 224, 1344, 1)])
@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("variant", [184, 185, 186])
def test_wsreg_expand_gemm(B, H, Cin, Cout, act, variant, f16):
    """The pipelined weights-stationary GEMMs of the MBConv expand convolutions (variant 184: K <= 224, two workgroups of four
    waves per CU; 185 / 186: K = 384, four / eight waves), bf16 and fp16 operands: against torch on the same rounded operands, and
    BIT-identical to the tile kernel (variant 131 / 132) -- same k order, same epilogue arithmetic. Shapes: ragged last tile, a
    single partial tile, several tiles per workgroup, every K, channel counts that are not a multiple of the 128-channel slice
    (1344, 1152, 192)."""
    from isbfsar_amd.hpe_engine import f16_to_f32, f32_to_f16
    if (variant == 184) == (Cin == 384):
        pytest.skip("variant 184 is built for K <= 224, variants 185 / 186 for K = 384")
    if not act:
        pytest.skip("the product build holds the SiLU forms (every expand convolution has one)")
    rng = np.random.default_rng(B * 1000 + Cin)
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    cvt, back = (f32_to_f16, f16_to_f32) if f16 else (f32_to_bf16, bf16_to_f32)
    out, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, act, None, None, variant=variant, f16=f16)
    tile, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, act, None, None, variant=131 if Cout % 192 == 0 else 132, f16=f16)
    assert np.array_equal(out, tile)
    ref = _ref16(x, w, scale, shift, 1, 1, act, None, "f16" if f16 else "bf16")
    got = back(out)
    assert np.abs(got - ref).max() <= (2.0 ** -10 if f16 else 2.0 ** -7) * max(1.0, np.abs(ref).max()) + 1e-4
    out2, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, act, None, None, variant=variant, f16=f16, iters=3)
    assert np.array_equal(out2, out)


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,H,Cin,Cout", [(8, 32, 96, 384), (40, 16, 192, 768), (130, 8, 384, 768)])
def test_auto_dispatch_of_a_short_k_gemm_without_activation(B, H, Cin, Cout, f16):
    """a legal auto-dispatched shape -- 1x1, Cin 96 / 192 / 384, NO activation, no residual or gate, M >= 8192 --
    must not be sent to the weights-stationary variants, whose product build holds the SiLU forms only (launch_conv_ws refuses the
    rest): variant 0 has to run it on the tile kernels (131 / 132), with the tile kernels' bits."""
    from isbfsar_amd.hpe_engine import f16_to_f32, f32_to_f16
    rng = np.random.default_rng(B + Cin)
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    cvt, back = (f32_to_f16, f16_to_f32) if f16 else (f32_to_bf16, bf16_to_f32)
    out, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, 0, None, None, variant=0, f16=f16)
    tile, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, 0, None, None, variant=131 if Cout % 192 == 0 else 132, f16=f16)
    assert np.array_equal(out, tile)
    ref = _ref16(x, w, scale, shift, 1, 1, 0, None, "f16" if f16 else "bf16")
    assert np.abs(back(out) - ref).max() <= (2.0 ** -10 if f16 else 2.0 ** -7) * max(1.0, np.abs(ref).max()) + 1e-4


@pytest.mark.parametrize("B,H,Cin,Cout", [(64, 16, 1344, 224), (70, 16, 1152, 224), (65, 16, 768, 192), (3, 16, 768, 192), (130, 16, 1344, 224)])
def test_gated_projection_with_stationary_weights(B, H, Cin, Cout):
    """gemm1x1_wsk_kernel (conv_wsk.hip, tile variant 157; round 5): the SE-gated fp16 projections of the 16 x 16 stage with a wave's
    weights for all of K in registers and the gate applied once per CU on the activations' way into LDS -- same k order, same gate
    rounding (f16(f32(x) * g)), same shared epilogue as the tile kernel: bit-identical to variant 143 / 141; and against torch on
    the same rounded operands. Shapes: one tile sequence longer than the others (ragged tile count), fewer tiles than sequences.
    MEASURED 2x SLOWER than the tile kernels (EXPERIMENTS.md round 5: a lone wave per SIMD, 32 KiB in flight per CU), so the kernel is
    compiled in probe builds only; the gate's single rounding (v_fma_mix) differs from the tile kernel's mul + convert in rare
    double-rounding cases, hence one 16-bit step of tolerance."""
    import ctypes
    from isbfsar_amd import _lib
    from isbfsar_amd.hpe_engine import f16_to_f32, f32_to_f16
    if not ctypes.CDLL(_lib.LIB_PATH).isbfsar_probe_build():
        pytest.skip("variant 157 is compiled in probe builds only (ISB_BUILD_PROBES=1 python -m isbfsar_amd.build --force)")
    rng = np.random.default_rng(B + Cin)
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    res = f32_to_f16(rng.normal(0, 1, (B, H, H, Cout)).astype(np.float32))
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32)
    out, _ = conv_debug(f32_to_f16(x), w, scale, shift, 1, 1, 0, res, gate, variant=157, f16=True)
    tile, _ = conv_debug(f32_to_f16(x), w, scale, shift, 1, 1, 0, res, gate, variant=143 if Cout == 224 else 141, f16=True)
    fo, ft = f16_to_f32(out), f16_to_f32(tile)
    assert np.all(np.abs(fo - ft) <= 2.0 ** -10 * np.abs(ft) + 2e-6) and np.mean(out != tile) < 1e-3
    again, _ = conv_debug(f32_to_f16(x), w, scale, shift, 1, 1, 0, res, gate, variant=157, f16=True, iters=2)
    assert np.array_equal(out, again)
    assert np.isfinite(f16_to_f32(out)).all() and float(np.abs(f16_to_f32(out)).max()) > 0


def _gemm_ref(A, W, bias, a_bias, a_add, act, a_act):
    """float64 restatement of gemm_f32.hip's contract."""
    actf = {0: lambda v: v, 1: lambda v: np.maximum(v, 0), 2: lambda v: v / (1 + np.exp(-v)), 3: lambda v: 1 / (1 + np.exp(-v))}
    a = A.astype(np.float64).sum(axis=0)
    if a_bias is not None:
        a = a + a_bias
    a = actf[a_act](a)
    if a_add is not None:
        a = a + a_add[np.arange(a.shape[0]) % a_add.shape[0]]
    c = a @ W.astype(np.float64).T
    if bias is not None:
        c = c + bias
    return actf[act](c)


@pytest.mark.parametrize("M,N,K,opts", [
    (300, 288, 1280, {}),                                           # the pose head's shape: 128 x 96 tiles, 16-byte staging
    (257, 732, 366, {"bias": True, "act": 1}),                      # MLP fc1: rows only 8-byte aligned, K tail of 14
    (190, 256, 732, {"bias": True}),                                # MLP fc2
    (95, 512, 256, {"a_add": 30}),                                  # tuple projections: positional table added on load
    (64, 256, 13050, {"splits": 6, "bias": True, "act": 1}),        # discriminator fc1: split-K, odd-ish K
    (33, 64, 256, {"a_parts": 3, "a_bias": True, "a_act": 1}),      # partial sums + bias + ReLU on load (N <= 64 tiles)
    (700, 1, 64, {"bias": True, "act": 3}),                         # N = 1 (N <= 32 tiles)
    (130, 130, 37, {"bias": True}),                                 # K not a multiple of 2: scalar staging, K tail of 5
    (200, 160, 96, {"a_offset": 2}),                                # A shifted to an 8-byte boundary
    (200, 160, 96, {"a_offset": 1, "act": 2}),                      # ... and to a 4-byte boundary
])
def test_gemm_f32(M, N, K, opts):
    """The exact-f32 Linear kernel against float64 on every staging path (16 / 8 / 4-byte loads, K tails, A-operand
    transforms, split-K) and tile shape. f32 products summed in f32: a few ulp of the accumulated magnitude."""
    from isbfsar_amd.hpe_engine import gemm_f32_debug
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    parts = opts.get("a_parts", 1)
    A = rng.normal(0, 1, (parts, M, K)).astype(np.float32)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.normal(0, 0.5, N).astype(np.float32) if opts.get("bias") else None
    a_bias = rng.normal(0, 0.5, K).astype(np.float32) if opts.get("a_bias") else None
    a_add = rng.normal(0, 0.1, (opts["a_add"], K)).astype(np.float32) if opts.get("a_add") else None
    act, a_act = opts.get("act", 0), opts.get("a_act", 0)
    out, _ = gemm_f32_debug(A, W, bias, a_bias, a_add, act=act, a_act=a_act, splits=opts.get("splits", 1),
                            a_offset=opts.get("a_offset", 0))
    ref = _gemm_ref(A, W, bias, a_bias, a_add, act, a_act)
    scale = np.sqrt(parts) * (1.0 + (0.5 if a_bias is not None else 0.0))
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, scale) * np.sqrt(K / 256 + 1)
    out2, _ = gemm_f32_debug(A, W, bias, a_bias, a_add, act=act, a_act=a_act, splits=opts.get("splits", 1),
                             a_offset=opts.get("a_offset", 0), iters=2)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("H,Cin,Cout,variant", [(16, 224, 1344, 184), (8, 384, 2304, 186), (32, 96, 384, 184)])
def test_pipelined_expand_at_full_size(H, Cin, Cout, variant):
    """BASELINE size (256 frames per launch: 512 workgroups x 11 - 24 tiles): the pipelined weights-stationary kernels are
    bit-identical to the tile kernel they replace, launch after launch."""
    rng = np.random.default_rng(H + Cin)
    x = f32_to_bf16(rng.normal(0, 1, (256, H, H, Cin)).astype(np.float32))
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    out, _ = conv_debug(x, w, scale, shift, 1, 1, 1, None, None, variant=variant, iters=3)
    tile, _ = conv_debug(x, w, scale, shift, 1, 1, 1, None, None, variant=131)
    assert np.array_equal(out, tile)
    auto, _ = conv_debug(x, w, scale, shift, 1, 1, 1, None, None, variant=0)     # what the network launches for this layer
    assert np.array_equal(auto, tile)


def _ref_f16(x, w, scale, shift, act, res, gate):
    """torch-CPU reference of a 1x1 convolution on fp16-rounded operands (ConvArgs.f16)"""
    xb = torch.from_numpy(x).half().float()
    if gate is not None:
        xb = (xb * torch.from_numpy(gate)[:, None, None, :]).half().float()
    wf = (torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1, 1)).half().float()[:, 0, 0, :]
    y = xb.double() @ wf.double().T + torch.from_numpy(shift).double()
    if act:
        y = y * torch.sigmoid(y)
    if res is not None:
        y = y + torch.from_numpy(res).half().double()
    return y.numpy()


F16_CASES = [
    # B, H, Cin, Cout, act, res, gate          the shapes of the two 8x8 stages + the 640 -> 1280 convolution
    (40, 8, 384, 2304, 1, False, False),       # expand (weights-stationary 185 at this M, tile kernel 131 as an explicit variant)
    (3, 8, 640, 3840, 1, False, False),        # expand, small M
    (5, 8, 2304, 384, 0, True, True),          # SE-gated projection with residual
    (4, 8, 3840, 640, 0, True, True),
    (2, 8, 1344, 384, 0, False, True),         # the projection of the block that enters the fp16 stages
    (3, 8, 640, 1280, 1, False, False),        # 640 -> 1280 (the product stores f32 there; the bf16/fp16 store path is checked here)
    (1, 8, 2304, 384, 0, True, True),          # one frame
]


@pytest.mark.parametrize("variant", [0, 131, 132, 138, 141, 143, 144, 146, 147, 155, 156, 185, 186, 3147, 2138])
@pytest.mark.parametrize("case", F16_CASES)
def test_conv_f16_operands(case, variant):
    """fp16 operands (ConvArgs.f16, isb_hpe_cfg.precision 0: the 8x8 stages): x / weights / residual / output in IEEE
    fp16, f32 accumulate -- against an f64 reference on the same fp16-rounded operands; both sides round to fp16 once."""
    from isbfsar_amd.hpe_engine import f16_to_f32, f32_to_f16
    B, H, Cin, Cout, act, use_res, use_gate = case
    tv = variant % 1000
    if tv in (131, 132, 138, 185, 186) and use_gate:
        pytest.skip("un-gated kernels")
    if tv in (141, 143, 144, 146, 147, 155, 156) and not use_gate:
        pytest.skip("gated kernels")
    if tv in (185, 186) and (Cin != 384 or not act):
        pytest.skip("weights-stationary fp16 form: K = 384 with SiLU")
    if tv == 144 and Cout % 320 != 0:
        pytest.skip("128 x 320 tiles")
    if variant >= 2000 and (use_res is False and act):
        pass
    rng = np.random.default_rng(hash((case, 13)) % (2 ** 31))
    x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    res = rng.normal(0, 1, (B, H, H, Cout)).astype(np.float32) if use_res else None
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32) if use_gate else None
    out, ms = conv_debug(f32_to_f16(x), w, scale, shift, 1, 1, act, None if res is None else f32_to_f16(res), gate,
                         variant=variant, f16=True)
    got = f16_to_f32(out).astype(np.float64)
    ref = _ref_f16(x, w, scale, shift, act, res, gate)
    assert np.isfinite(got).all()
    # one fp16 rounding (2^-11 relative) + f32 accumulation order over K <= 3840
    tol = 2.0 ** -10 * np.maximum(0.25, np.abs(ref)) + 1e-4
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())


@pytest.mark.parametrize("form", [(1, True, True), (2, False, True), (2, True, True)])
def test_dwconv_f16_forms(form):
    """depthwise 3x3 + SiLU + pool in the fp16 forms (DwArgs.in_f16 / out_f16): fp16 -> fp16 at stride 1 and 2 (the blocks of
    fp16 stages) and stride 2 bf16 -> fp16 (the block that enters them under precision 3), against torch on the same rounded
    operands."""
    from isbfsar_amd.hpe_engine import dwconv_debug, f16_to_f32, f32_to_f16
    stride, in_f16, out_f16 = form
    B, H, Cc = 3, 8 * stride, 2304 if stride == 1 else 1344
    rng = np.random.default_rng(17 + stride)
    x = rng.normal(0, 1, (B, H, H, Cc)).astype(np.float32)
    w = (rng.normal(0, 1, (Cc, 3, 3)) / 3.0).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cc).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cc).astype(np.float32)
    xin = f32_to_f16(x) if in_f16 else f32_to_bf16(x)
    out, pooled, _ = dwconv_debug(xin, w, scale, shift, stride=stride, in_f16=in_f16, out_f16=out_f16)
    xr = torch.from_numpy(f16_to_f32(xin) if in_f16 else bf16_to_f32(xin)).permute(0, 3, 1, 2)
    wf = torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1)
    wf = (wf.half() if in_f16 else wf.bfloat16()).float().unsqueeze(1)
    if stride == 2:
        y = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wf, stride=2, groups=Cc)
    else:
        y = F.conv2d(xr, wf, padding=1, groups=Cc)
    y = y + torch.from_numpy(shift).view(1, -1, 1, 1)
    y = (y * torch.sigmoid(y)).permute(0, 2, 3, 1)
    ref = y.half().float().numpy()
    got = f16_to_f32(out)
    tol = 2.0 ** -10 * np.maximum(0.25, np.abs(ref))
    assert np.all(np.abs(got - ref) <= tol), float(np.abs(got - ref).max())
    np.testing.assert_allclose(pooled, got.reshape(B, -1, Cc).mean(axis=1), rtol=0, atol=2e-5)   # the pool sees the stored values


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("HW,Cc", [(8, 2304), (8, 3840), (8, 200), (16, 768), (16, 1344), (16, 200)])
def test_dwconv_map8_is_bit_identical(HW, Cc, f16):
    """8 x 8 and 16 x 16 maps run on dwconv3x3_map_kernel (slab staged in LDS once, zero ring for the padding): same taps in
    the same order as the general kernel -> the same bits, outputs and pooled means (200 channels: a partial last slab)."""
    from isbfsar_amd.hpe_engine import dwconv_debug, f32_to_f16
    rng = np.random.default_rng(Cc + int(f16) + HW)
    B = 5
    x = rng.normal(0, 1, (B, HW, HW, Cc)).astype(np.float32)
    w = (rng.normal(0, 1, (Cc, 3, 3)) / 3.0).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cc).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cc).astype(np.float32)
    xin = f32_to_f16(x) if f16 else f32_to_bf16(x)
    a, pa, _ = dwconv_debug(xin, w, scale, shift, stride=1, in_f16=f16, out_f16=f16, general=2)
    g, pg, _ = dwconv_debug(xin, w, scale, shift, stride=1, in_f16=f16, out_f16=f16, general=1)
    assert np.array_equal(a, g) and np.array_equal(pa, pg)
    assert np.abs(pa).max() > 0


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,HW,Cc", [(5, 8, 2304), (3, 8, 3840), (2, 8, 128), (5, 16, 768), (3, 16, 1344), (2, 16, 1152), (1, 16, 64)])
def test_dwconv_taps_on_the_matrix_pipe(B, HW, Cc, f16):
    """dwconv3x3_mm_kernel (round 5; DwArgs.general = 3: the stand-alone partner of the fused 8 x 8 front): the nine taps as three Toeplitz-band
    v_mfma_f32_16x16x32 per 8 channels x 32 pixels instead of 288 v_dot2 (dw_mm.h). Products of 16-bit values are exact in f32
    and the sums are f32 in both forms, so against the v_dot2 kernel the results may differ by the summation ORDER only: the
    16-bit outputs agree except at rounding boundaries (one ulp of the storage type there), pooled means to f32 rounding; and
    against torch on the same rounded operands like every depthwise kernel."""
    from isbfsar_amd.hpe_engine import dwconv_debug, f16_to_f32, f32_to_f16
    rng = np.random.default_rng(Cc + int(f16) + HW + B)
    x = rng.normal(0, 1, (B, HW, HW, Cc)).astype(np.float32)
    w = (rng.normal(0, 1, (Cc, 3, 3)) / 3.0).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cc).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cc).astype(np.float32)
    cvt, back = (f32_to_f16, f16_to_f32) if f16 else (f32_to_bf16, bf16_to_f32)
    xin = cvt(x)
    a, pa, _ = dwconv_debug(xin, w, scale, shift, stride=1, in_f16=f16, out_f16=f16, general=3)    # the matrix-pipe kernel
    g, pg, _ = dwconv_debug(xin, w, scale, shift, stride=1, in_f16=f16, out_f16=f16, general=2)    # v_dot2 taps
    fa, fg = back(a), back(g)
    ulp = (2.0 ** -10 if f16 else 2.0 ** -7) * np.abs(fg) + 2e-6            # one step of the storage type, or f32 rounding of a nine-term sum near zero
    assert np.all(np.abs(fa - fg) <= ulp), float(np.abs(fa - fg).max())
    assert np.mean(a != g) < 5e-3, float(np.mean(a != g))
    np.testing.assert_allclose(pa, pg, rtol=0, atol=2e-5)
    np.testing.assert_allclose(pa, fa.reshape(B, -1, Cc).mean(axis=1), rtol=0, atol=2e-5)          # the pool sees the stored values
    xr = torch.from_numpy(back(xin)).permute(0, 3, 1, 2)
    wf = torch.from_numpy(w) * torch.from_numpy(scale).view(-1, 1, 1)
    wf = (wf.half() if f16 else wf.bfloat16()).float().unsqueeze(1)
    y = F.conv2d(xr, wf, padding=1, groups=Cc) + torch.from_numpy(shift).view(1, -1, 1, 1)
    y = (y * torch.sigmoid(y)).permute(0, 2, 3, 1)
    ref = (y.half() if f16 else y.bfloat16()).float().numpy()
    tol = (2.0 ** -10 if f16 else 2.0 ** -7) * np.maximum(0.25 if f16 else 1.0, np.abs(ref))
    assert np.all(np.abs(fa - ref) <= tol), float(np.abs(fa - ref).max())
    # a repeated launch gives the same bits (fixed summation orders, no atomics)
    a2, pa2, _ = dwconv_debug(xin, w, scale, shift, stride=1, in_f16=f16, out_f16=f16, general=3, iters=2)
    assert np.array_equal(a, a2) and np.array_equal(pa, pa2)


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("HW,Cc,cse", [(8, 2304, 96), (8, 3840, 160), (16, 768, 48), (16, 1344, 56), (16, 200, 13)])
def test_dwconv_map_fc1_fold(HW, Cc, cse, f16):
    """Squeeze-excite FC1 folded into the depthwise launch (DwArgs.se_w1; the single-frame path runs this way): each channel
    slab's workgroup leaves its share of pooled @ W1^T. The LDS-map kernel's partials are the general kernel's bits, and the
    slabs add up to the f32 product of the pooled means."""
    from isbfsar_amd.hpe_engine import dwconv_fc1_debug, f32_to_f16
    rng = np.random.default_rng(Cc + cse + int(f16))
    B = 3
    x = rng.normal(0, 1, (B, HW, HW, Cc)).astype(np.float32)
    w = (rng.normal(0, 1, (Cc, 3, 3)) / 3.0).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cc).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cc).astype(np.float32)
    w1 = (rng.normal(0, 1, (cse, Cc)) / np.sqrt(Cc)).astype(np.float32)
    xin = f32_to_f16(x) if f16 else f32_to_bf16(x)
    a, pa, parts_a = dwconv_fc1_debug(xin, w, scale, shift, w1, in_f16=f16, out_f16=f16)
    g, pg, parts_g = dwconv_fc1_debug(xin, w, scale, shift, w1, in_f16=f16, out_f16=f16, general=True)
    assert np.array_equal(a, g) and np.array_equal(pa, pg)
    assert parts_a.shape == parts_g.shape == (-(-Cc // (128 if HW == 8 else 64)), B, cse)
    assert np.array_equal(parts_a, parts_g)
    want = pa.astype(np.float64) @ w1.astype(np.float64).T
    np.testing.assert_allclose(parts_a.astype(np.float64).sum(axis=0), want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("shape", [(9, 2304, 384, 155), (5, 3840, 640, 156), (3, 1344, 384, 155), (2, 2304, 640, 156), (1, 768, 192, 155)])
def test_gated_projection_with_loader_waves_is_bit_identical(shape, f16):
    """gemm1x1_lw_kernel (variants 155 / 156: 128-row tiles, 8 consumer + 4 loader waves, one barrier per k-step) against the
    tile kernel 146 / 144 on the 8 x 8 stages' projections: same LDS images, gate arithmetic and k order -> the same bits
    (ragged M: 9 / 5 / 3 frames of 64 rows do not fill the last 128-row tile)."""
    from isbfsar_amd.hpe_engine import f32_to_f16
    B, Cin, Cout, v = shape
    rng = np.random.default_rng(Cin + Cout + B)
    x = rng.normal(0, 1, (B, 8, 8, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.8, 1.2, Cout).astype(np.float32)
    shift = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    res = rng.normal(0, 1, (B, 8, 8, Cout)).astype(np.float32)
    gate = rng.uniform(0.1, 0.9, (B, Cin)).astype(np.float32)
    cvt = f32_to_f16 if f16 else f32_to_bf16
    a, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, 0, cvt(res), gate, variant=v, f16=f16)
    ref_v = 146 if Cout % 192 == 0 else 144
    b, _ = conv_debug(cvt(x), w, scale, shift, 1, 1, 0, cvt(res), gate, variant=ref_v, f16=f16)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["g1_131", "g1_138", "gate_146", "gate_155", "ws_184", "ws_186", "c3_161", "c3_171", "c3_167", "fused", "dw1", "dw1mm", "dw2", "splitk"])
def test_f16_storage_saturates_instead_of_overflowing(kind):
    """fp16 storage must never produce an inf (it would poison every later layer): the kernels templated on the storage type set
    MODE.FP16_OVFL at their start, so conversions clamp to +-65504 (conv_common.h T16::enter). Every fp16 kernel family is driven
    past the fp16 range here: finite outputs, the largest exactly 65504, and in-range values untouched."""
    from isbfsar_amd.hpe_engine import dwconv_debug, f16_to_f32, f32_to_f16, fused_mb_debug
    rng = np.random.default_rng(len(kind))
    big = 3.0e4

    def check(out):
        got = f16_to_f32(out)
        assert np.isfinite(got).all()
        assert np.abs(got).max() == 65504.0
        assert (np.abs(got) < 6.0e4).mean() > 0.2            # not everything saturated: the clamp is per value

    if kind.startswith(("g1", "gate", "ws", "splitk")):
        B, H, Cin, Cout, gate, act = {"g1_131": (9, 8, 160, 384, False, 0), "g1_138": (1, 8, 160, 128, False, 0), "gate_146": (5, 8, 256, 384, True, 0),
                                      "gate_155": (70, 8, 256, 384, True, 0), "ws_184": (40, 16, 224, 256, False, 1),
                                      "ws_186": (300, 8, 384, 256, False, 1), "splitk": (1, 8, 512, 128, True, 0)}[kind]
        x = rng.normal(0, 1, (B, H, H, Cin)).astype(np.float32)
        w = (rng.normal(0, 1, (Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32)
        g = rng.uniform(0.5, 1.0, (B, Cin)).astype(np.float32) if gate else None
        v = {"g1_131": 131, "g1_138": 138, "gate_146": 146, "gate_155": 155, "ws_184": 184, "ws_186": 186, "splitk": 4147}[kind]
        out, _ = conv_debug(f32_to_f16(x), w, np.full(Cout, big, np.float32), np.zeros(Cout, np.float32), 1, 1, act, None, g, variant=v, f16=True)
        check(out)
    elif kind.startswith("c3"):
        B, H, W, Cin, Cout, v = {"c3_161": (2, 16, 16, 64, 192, 161), "c3_171": (1, 8, 128, 32, 32, 171), "c3_167": (1, 32, 32, 96, 192, 167)}[kind]
        x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
        w = (rng.normal(0, 1, (Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
        out, _ = conv_debug(f32_to_f16(x), w, np.full(Cout, big, np.float32), np.zeros(Cout, np.float32), 3, 1, 1, None, None, variant=v, f16=True)
        check(out)
    elif kind == "fused":
        x = rng.normal(0, 1, (1, 64, 64, 64)).astype(np.float32)
        w1 = (rng.normal(0, 1, (256, 3, 3, 64)) / 24.0).astype(np.float32)
        w2 = (rng.normal(0, 1, (64, 256)) / 16.0).astype(np.float32)
        one, zero = np.ones(256, np.float32), np.zeros(256, np.float32)
        out, _ = fused_mb_debug(f32_to_f16(x), w1, 300.0 * one, zero, w2, 300.0 * one[:64], zero[:64], None, 1, f16=True)   # the E tile saturates too
        check(out)
    else:
        stride = 1 if kind in ("dw1", "dw1mm") else 2
        C_ = 256
        x = rng.normal(0, 1, (2, 8 * stride, 8 * stride, C_)).astype(np.float32)
        w = (rng.normal(0, 1, (C_, 3, 3)) / 3.0).astype(np.float32)
        out, pooled, _ = dwconv_debug(f32_to_f16(x), w, np.full(C_, big, np.float32), np.zeros(C_, np.float32), stride=stride, in_f16=True, out_f16=True,
                                      general=3 if kind == "dw1mm" else 0)
        check(out)
        assert np.isfinite(pooled).all()


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,cin,cexp", [(3, 224, 1344), (37, 224, 1344), (64, 192, 1152), (21, 192, 768), (130, 224, 1344)])
def test_fused_front_16x16_forms_are_bit_identical(B, cin, cexp, f16):
    """The front half of a stride-1 MBConv block on 16 x 16 maps (expand + SiLU -> depthwise + SiLU -> D + pool) three ways: two launches
    (expand GEMM + matrix-pipe depthwise kernel), mbfront16_kernel (round 5: every wave does everything) and mbfront16r_kernel (round
    6: producer / consumer waves, four per SIMD, units cut into ranges that may cross a slice boundary -- the batch sizes here give
    ranges of one, several and a fractional number of samples per workgroup, XCDs with fewer samples than others, and 1344 = 10.5
    slices). Same arithmetic and summation orders: D and the pooled means bit for bit."""
    from isbfsar_amd.hpe_engine import f32_to_f16, mbfront_debug
    rng = np.random.default_rng(B * 7 + cin + cexp + int(f16))
    x = rng.normal(0, 1, (B, 16, 16, cin)).astype(np.float32)
    w1 = (rng.normal(0, 1, (cexp, cin)) / np.sqrt(cin)).astype(np.float32)
    s1 = rng.uniform(0.8, 1.2, cexp).astype(np.float32)
    b1 = rng.uniform(-0.2, 0.2, cexp).astype(np.float32)
    dww = (rng.normal(0, 1, (cexp, 3, 3)) / 3.0).astype(np.float32)
    s2 = rng.uniform(0.8, 1.2, cexp).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, cexp).astype(np.float32)
    x16 = f32_to_f16(x) if f16 else f32_to_bf16(x)
    d0, p0, _ = mbfront_debug(x16, w1, s1, b1, dww, s2, b2, f16=f16, form=0)
    assert np.abs(p0).max() > 0 and np.isfinite(p0).all()
    for form in (1, 2):
        d, pl, _ = mbfront_debug(x16, w1, s1, b1, dww, s2, b2, f16=f16, form=form, iters=2)
        assert np.array_equal(d, d0), (form, float(np.mean(d != d0)))
        assert np.array_equal(pl, p0), (form, float(np.abs(pl - p0).max()))


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B", [3, 32, 45, 130])
def test_fused_front_8x8_forms_are_bit_identical(B, f16):
    """The same for the 384 -> 2304 blocks of the 8 x 8 stage: two launches, mbfront8_kernel (round 4 / 5) and mbfront8r_kernel (round 6:
    four producer and eight consumer waves per workgroup, one workgroup per CU; ranges of 2304 / 128 = 18 slices x an XCD's samples that
    cross slice boundaries): D and the pooled means bit for bit."""
    from isbfsar_amd.hpe_engine import f32_to_f16, mbfront_debug
    cin, cexp = 384, 2304
    rng = np.random.default_rng(B * 11 + int(f16))
    x = rng.normal(0, 1, (B, 8, 8, cin)).astype(np.float32)
    w1 = (rng.normal(0, 1, (cexp, cin)) / np.sqrt(cin)).astype(np.float32)
    s1 = rng.uniform(0.8, 1.2, cexp).astype(np.float32)
    b1 = rng.uniform(-0.2, 0.2, cexp).astype(np.float32)
    dww = (rng.normal(0, 1, (cexp, 3, 3)) / 3.0).astype(np.float32)
    s2 = rng.uniform(0.8, 1.2, cexp).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, cexp).astype(np.float32)
    x16 = f32_to_f16(x) if f16 else f32_to_bf16(x)
    d0, p0, _ = mbfront_debug(x16, w1, s1, b1, dww, s2, b2, f16=f16, form=0)
    assert np.abs(p0).max() > 0 and np.isfinite(p0).all()
    for form in (1, 2):
        d, pl, _ = mbfront_debug(x16, w1, s1, b1, dww, s2, b2, f16=f16, form=form, iters=2)
        assert np.array_equal(d, d0), (form, float(np.mean(d != d0)))
        assert np.array_equal(pl, p0), (form, float(np.abs(pl - p0).max()))


@pytest.mark.parametrize("B,C,cse", [(256, 2304, 96), (64, 768, 48), (33, 3840, 160), (9, 384, 24), (130, 1344, 56), (8, 2304, 96), (1, 1152, 48)])
def test_squeeze_excite_fcs_against_the_f64_definition_and_under_sharding(B, C, cse):
    """se_fc1_part_kernel + se_fc2_kernel on their own (isb_debug_se_fcs): gate = sigmoid(b2 + W2 silu(b1 + W1 pooled)) within f32 rounding
    of the f64 definition, and a sample's gate bit for bit the same whether it arrives alone, in a part of the batch or in all of it
    (both kernels sum in orders that depend on the layer alone; B <= 8 takes the prefetching FC2 form)."""
    from isbfsar_amd.hpe_engine import se_fcs_debug
    rng = np.random.default_rng(B * 7 + C + cse)
    pooled = rng.normal(0, 0.5, (B, C)).astype(np.float32)
    w1 = (rng.normal(0, 1, (cse, C)) / np.sqrt(C)).astype(np.float32)
    b1 = rng.uniform(-0.3, 0.3, cse).astype(np.float32)
    w2t = (rng.normal(0, 1, (cse, C)) / np.sqrt(cse)).astype(np.float32)
    b2 = rng.uniform(-0.5, 0.5, C).astype(np.float32)
    g, _ = se_fcs_debug(pooled, w1, b1, w2t, b2)
    mid = pooled.astype(np.float64) @ w1.T.astype(np.float64) + b1
    mid = mid / (1.0 + np.exp(-mid))
    want = 1.0 / (1.0 + np.exp(-(mid @ w2t.astype(np.float64) + b2)))
    assert np.abs(g - want).max() < 2e-5
    cut = max(1, B // 3)
    for lo, hi in ((0, cut), (cut, B), (B - 1, B)):
        if hi > lo:
            part, _ = se_fcs_debug(pooled[lo:hi], w1, b1, w2t, b2)
            assert np.array_equal(part, g[lo:hi]), (lo, hi)
