"""ResNet-50 trunk (hybrid input type) throughput by batch, device-resident images."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isbfsar_amd.rgb_engine import RgbEngine
from isbfsar_amd import resnet50
for B in (16, 64, 256):
    e = RgbEngine(device=0, max_batch=B)
    e.load_weights(resnet50.make_state(0))
    x = torch.rand((B, 3, 224, 224), device="cuda")
    for _ in range(3): e.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n): e.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt * 1e3:.2f} ms, {B / dt:.0f} images/s, {2 * resnet50.macs_per_image() * B / dt / 1e12:.0f} TFLOP/s", flush=True)
    e.close()
