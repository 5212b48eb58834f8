"""CPU: known answers of the public efficientnetv2-l model asserted on the backbone oracle's OWN table
(oracle/effnetv2_oracle.py parses the published block strings; it imports nothing from the product package),
then the product's table (isbfsar_amd/effnetv2.py, mirrored in csrc/hpe_api.cpp kStages) is checked against it.

The backbone leg stays "parity unpinned" (no MetrABS source or weights in the reference tree); what
these tests pin is the ARCHITECTURE both sides implement:
  * 117,746,848 parameters without the classifier top, 117,234,272 of them trainable -- the totals Keras reports
    for EfficientNetV2L(include_top=False);
  * 15.99 GMAC per 256x256 crop including the 1280->288 pose head;
  * [B,256,256,3] -> [B,8,8,1280] -> [B,8,8,288] (reference modules/hpe/setup/4_create_heads_onnx.py:10-19)."""
import numpy as np
import pytest

from oracle import effnetv2_oracle as eo


def test_block_string_parser():
    a = eo.parse_block_string("r10_k3_s2_e4_i96_o192_se0.25")
    assert (a["r"], a["k"], a["s"], a["e"], a["i"], a["o"], a["se"], a["c"]) == (10, 3, 2, 4, 96, 192, 0.25, 0)
    assert eo.parse_block_string("r4_k3_s1_e1_i32_o32_c1")["c"] == 1
    with pytest.raises(ValueError):
        eo.parse_block_string("r4_k3_s1_e1_i32")
    with pytest.raises(ValueError):
        eo.parse_block_string("r4_k3_s1_e1_i32_o32_?")


def test_known_answers_of_the_public_model():
    blocks = eo.oracle_blocks()
    assert len(blocks) == 4 + 7 + 7 + 10 + 19 + 25 + 7 == 79
    n = eo.count_parameters(blocks)
    assert n["total"] == 117_746_848 and n["trainable"] == 117_234_272
    macs = eo.count_macs(blocks)
    assert abs(macs / 1e9 - 15.99) < 0.005, macs
    assert blocks[0].in_hw == 128 and blocks[-1].out_hw == 8 and blocks[-1].cout == 640
    # squeeze widths follow the block INPUT: 96*0.25 = 24 for the first MBConv, then 48, 56, 96, 160
    assert sorted({b.cse for b in blocks if b.kind == "mb"}) == [24, 48, 56, 96, 160]
    # identity skips: every repeat after the first of a stage, plus the whole stride-1 32->32 stage
    assert sum(b.residual for b in blocks) == 79 - 6


def test_product_table_equals_the_public_one():
    from isbfsar_amd import effnetv2 as prod
    pb, ob = prod.blocks(), eo.oracle_blocks()
    assert len(pb) == len(ob)
    for p, o in zip(pb, ob):
        assert (p.idx, p.kind, p.cin, p.cout, p.cexp, p.stride, p.cse, p.residual, p.in_hw, p.out_hw) == \
               (o.idx, o.kind, o.cin, o.cout, o.cexp, o.stride, o.cse, o.residual, o.in_hw, o.out_hw)
    assert prod.macs_per_crop() == eo.count_macs()
    # the weight container carries exactly the public model's convolution / SE tensors (BatchNorm folded to
    # scale + shift = 2 values per channel, where the public count has 4) + the pose head
    shapes = prod.tensor_shapes()
    n_w = sum(int(np.prod(s)) for k, s in shapes.items() if not k.startswith("head.") and k.split(".")[-1] not in ("scale", "shift"))
    n_bn = sum(int(np.prod(s)) for k, s in shapes.items() if k.split(".")[-1] == "scale")
    assert n_w + 4 * n_bn == eo.count_parameters()["total"]
    assert shapes["head.weight"] == (288, 1280)


def test_forward_shape_contract():
    """a 64x64 crop through the f32 definition: /32 down-sampling and 1280 features (the 256x256 contract, 8x8x1280,
    is exercised on the GPU box in tests/test_hpe_gpu.py; a full-size CPU pass takes ~1 s per crop)"""
    from isbfsar_amd import effnetv2 as prod
    net = eo.EffNetV2LOracle(prod.make_state(0), "f32")
    x = np.random.default_rng(0).random((1, 64, 64, 3), dtype=np.float32)
    f = net.backbone(x)
    assert f.shape == (1, 2, 2, 1280) and np.isfinite(f).all()
    assert net.head(f).shape == (1, 2, 2, 288)
