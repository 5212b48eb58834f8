"""CPU: the YOLOv4 detector's layer table (isbfsar_amd/yolov4.py) against known answers of the public model and against
the plan compiled into the library (isb_det_describe needs no GPU), and the oracle's pre-processing / decode contracts."""
import numpy as np

from isbfsar_amd import yolov4 as Y


def test_known_answers_of_the_public_model():
    L = Y.conv_layers()
    assert len(L) == 110                                   # 107 Conv-BN-activation blocks + 3 detection convs
    n_w = sum(l.cout * l.k * l.k * l.cin for l in L)
    n_bn = sum(4 * l.cout for l in L if l.bn)
    n_bias = sum(l.cout for l in L if not l.bn)
    assert n_w + n_bn + n_bias == 64_429_405               # what `sum(p.numel())` + BatchNorm buffers gives for Yolov4(n_classes=80)
    # 128.46 BFLOPs at 608 x 608 (darknet's own count for yolov4.cfg) scale with the area: 22.8 GFLOP at 256 x 256
    assert abs(2 * Y.macs_per_frame() / 1e9 - 128.46 * (256 / 608) ** 2) < 0.15
    assert Y.N_BOXES == 4032                               # hpe.py:60
    assert [l.cout for l in L if not l.bn] == [255, 255, 255]


def test_library_plan_matches_the_table():
    from isbfsar_amd.build import build
    build(verbose=False)
    from isbfsar_amd.det_engine import describe_convs
    lib = describe_convs()
    tab = Y.conv_layers()
    assert len(lib) == len(tab)
    for (name, cin, cout, k, stride, act, bn), l in zip(lib, tab):
        assert (name, cin, cout, k, stride, act, bn) == (l.name, l.cin, l.cout, l.k, l.stride, l.act, l.bn)


def test_oracle_preprocess_and_decode_contracts():
    from oracle.yolov4_oracle import YoloV4Oracle, area_resize_u8, preprocess
    f = np.zeros((480, 640, 3), np.uint8)
    f[:, :, 0], f[:, :, 2] = 10, 200                       # BGR
    img = preprocess(f)
    assert img.shape == (256, 256, 3) and np.allclose(img[..., 0], 200 / 255) and np.allclose(img[..., 2], 10 / 255)
    g = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    a = area_resize_u8(g)
    # an output pixel is the mean of a 1.875 x 2.5 source rectangle: the resized image keeps the mean, loses variance
    assert abs(float(a.mean()) - float(g.mean())) < 0.5 and a.std() < 0.6 * g.std()
    # decode of all-zero maps: sigmoid(0) = 0.5 -> box centres at (cell + 0.5) / grid, confs = 0.25, 4032 boxes, scale order 8/16/32
    maps = [np.zeros((1, hw, hw, 255), np.float32) for hw in (32, 16, 8)]
    boxes, confs = YoloV4Oracle.decode(maps)
    assert boxes.shape == (1, 4032, 1, 4) and confs.shape == (1, 4032, 80) and np.allclose(confs, 0.25)
    cx = (boxes[0, :, 0, 0] + boxes[0, :, 0, 2]) / 2
    assert np.allclose(cx[:32], (np.arange(32) + 0.5) / 32, atol=1e-6)
    w0 = boxes[0, 0, 0, 2] - boxes[0, 0, 0, 0]
    assert np.isclose(w0, 12 / 8 / 32)                     # anchor 12 px wide on the stride-8 map, normalised by the grid
