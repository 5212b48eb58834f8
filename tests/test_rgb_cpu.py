"""CPU: the ResNet-50 trunk's description (isbfsar_amd/resnet50.py) and its oracle (oracle/resnet50_oracle.py, parity
unpinned) against the published figures of torchvision's resnet50, and the checkpoint key mapping of the hybrid branch."""
import numpy as np

from isbfsar_amd import resnet50
from oracle import resnet50_oracle as ro


def test_known_answers_of_the_public_architecture():
    # torchvision.models.resnet50: 25,557,032 parameters, of which the fc layer 2048 * 1000 + 1000
    assert ro.count_parameters() == 25_557_032 - 2_049_000 == 23_508_032
    assert resnet50.count_parameters() == ro.count_parameters()
    # 4.09 GMAC per 224 x 224 image in torchvision's model table (with the fc's 2.05 M)
    assert abs(ro.count_macs() + 2_048_000 - 4.09e9) < 0.005e9
    assert resnet50.macs_per_image() == ro.count_macs()
    ob, pb = ro.oracle_blocks(), resnet50.blocks()
    assert len(ob) == len(pb) == 16 and sum(b["down"] for b in ob) == 4
    for o, p in zip(ob, pb):
        assert (o["cin"], o["planes"], o["stride"], o["in_hw"], o["out_hw"], o["down"]) == (p.cin, p.planes, p.stride, p.in_hw, p.out_hw, p.downsample)
    assert ob[-1]["out_hw"] == 7 and ob[-1]["planes"] * 4 == 2048


def test_trunk_key_mapping_round_trip():
    """resnet50.state_from_torch: the keys of TRXOS's nn.Sequential trunk (features_extractor.rgb.<child index>...; model.py:274)
    and of a plain torchvision state_dict -> blob tensors (OIHW -> OHWI, BatchNorm folded)."""
    st = resnet50.make_state(2)
    for prefix in ("features_extractor.rgb.", ""):
        sd = resnet50.to_torch_state(st, prefix)
        if prefix:
            assert "features_extractor.rgb.0.weight" in sd and "features_extractor.rgb.4.0.downsample.0.weight" in sd
            assert "features_extractor.rgb.7.2.bn3.running_var" in sd
        else:
            assert "conv1.weight" in sd and "layer4.2.conv3.weight" in sd
        back = resnet50.state_from_torch(sd, prefix)
        assert list(back) == list(st)
        for k in st:
            np.testing.assert_allclose(back[k], st[k], rtol=3e-7, atol=0)


def test_oracle_runs_and_modes_agree():
    st = resnet50.make_state(0)
    x = np.random.default_rng(0).normal(0, 1, (1, 3, 224, 224)).astype(np.float32)
    f32 = ro.ResNet50Oracle(st, "f32").forward(x)
    b16 = ro.ResNet50Oracle(st, "bf16").forward(x)
    assert f32.shape == (1, 2048) and np.isfinite(f32).all() and f32.std() > 1e-3
    assert np.linalg.norm(b16 - f32) / np.linalg.norm(f32) < 5e-2
