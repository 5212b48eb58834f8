"""bench.py workloads that include the pose stage (bench support, not product code: the
cpu_baseline legs import oracle/).

pipeline : BASELINE.json configs[3] per-GPU shard -- 256 synthetic 640x480 frames per GPU (8 cameras
           x 32 time steps) -> HPE (122 joints) -> 30-deep ring per camera -> 256 skeleton windows ->
           AR embed + 60-class tuple cross-attention match + open-set score (-> all-gather when N>1)
hpe      : BASELINE.json configs[1] -- 256 frames, pose stage only
"""
from __future__ import annotations

import contextlib
import json
import os
import time

import numpy as np

from isbfsar_amd import effnetv2, synth, weights
from isbfsar_amd.dist import all_gather_records, pack_records
from isbfsar_amd.engine import ArEngine
from isbfsar_amd.hpe_engine import HpeEngine, pose_windows

MFMA_PEAK_TFLOPS_BF16 = 2500.0     # MI355X_MICROARCH.md, dense
_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "isbfsar_amd", "assets")


def make_gather(args, world, dev):
    """(RecordGather | None, force): N > 1 on RCCL -> the library-owned communicator (isb_dist_*); the gloo rehearsal on a
    one-GPU box keeps torch.distributed as the test double. force = the one-rank rehearsal of the N > 1 branch."""
    force = bool(getattr(args, "force_dist", False))
    if world > 1 and getattr(args, "dist_backend", "nccl") == "nccl" and os.environ.get("ISB_BENCH_TORCH_GATHER") != "1":
        from isbfsar_amd.dist import RecordGather
        return RecordGather(dev), force
    return None, force


def usable_cores() -> int:
    """CPU threads this process may actually use (affinity mask and cgroup quota), not the host's."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def latest_traffic_json():
    """profiles/rNN_traffic.json of the latest round (PMC FETCH_SIZE / WRITE_SIZE passes, tools/round_profiles.sh)"""
    import glob
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_traffic.json")))
    return found[-1] if found else None


def latest_power(workload: str, achieved=None, peak=None):
    """profiles/rNN_power_<workload>.txt of the latest round (tools/power_probe.sh: rocm-smi samples beside a 400-600-step run of the
    workload) as the roofline line's `power` object, or None. The pose and pipeline steps run AT the package power cap: the clock the
    governor leaves is below the 2.4 GHz `peak` is priced at, so the object also carries achieved / (peak x sclk / 2400)."""
    import glob
    import re
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r[0-9][0-9]_power_{workload}.txt")))
    if not found:
        return None
    try:
        text = open(found[-1]).read()
        m = re.search(r"median / max ([\d.]+) / ([\d.]+) / ([\d.]+) W; sclk as rocm-smi reports it (\d+) / (\d+) / (\d+) MHz", text)
        cap = re.search(r"package power cap ([\d.]+) W", text)
        if not m:
            return None
        out = {"what": "rocm-smi samples (0.2 s) beside a 400-600-step run of this workload in the round's profile session, not this run",
               "package_w_median": float(m.group(2)), "package_w_cap": float(cap.group(1)) if cap else None,
               "sclk_mhz_median": int(m.group(5)), "source": "profiles/" + os.path.basename(found[-1])}
        if achieved and peak:
            out["frac_at_that_clock"] = round(achieved / (peak * int(m.group(5)) / 2400.0), 4)
        return out
    except (OSError, ValueError):
        return None


def median_time(fn, warm: int = 3, iters: int = 10):
    """BASELINE.md 4 protocol: `warm` untimed passes, then the MEDIAN wall time of `iters` timed ones.
    Returns (median seconds per pass, total seconds spent)."""
    t_all = time.perf_counter()
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), time.perf_counter() - t_all


def _softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def ar_parity(logits, is_true, embed, ref) -> dict:
    """GPU outputs vs the oracle's for the same windows: max-abs of logits / class probabilities / embedding, max-abs
    and L2 (root of the summed squares over the sample) of the open-set score."""
    logits, is_true = np.asarray(logits, np.float64), np.asarray(is_true, np.float64).reshape(-1)
    rt = np.asarray(ref["is_true"], np.float64).reshape(-1)
    out = {"n_windows": int(logits.shape[0]),
           "logits_maxabs": float(np.abs(logits - ref["logits"]).max()),
           "probs_maxabs": float(np.abs(_softmax(logits) - _softmax(np.asarray(ref["logits"], np.float64))).max()),
           "is_true_maxabs": float(np.abs(is_true - rt).max()),
           "is_true_l2": float(np.sqrt(((is_true - rt) ** 2).sum()))}
    if embed is not None:
        out["embed_maxabs"] = float(np.abs(np.asarray(embed, np.float64) - ref["query_features"]).max())
    return out


def igemm_macs_per_crop() -> int:
    """MACs executed by conv_igemm_kernel per crop (all 3x3/1x1 convs except the f32 stem)."""
    m = 0
    for b in effnetv2.blocks():
        o = b.out_hw * b.out_hw
        if b.kind == "fused":
            m += o * 9 * b.cin * b.cexp
            if b.cexp != b.cin:
                m += o * b.cexp * b.cout
        else:
            m += b.in_hw * b.in_hw * b.cin * b.cexp + o * b.cexp * b.cout
    return m + 64 * 640 * effnetv2.HEAD_OUT


def igemm_algorithmic_bytes(batch: int) -> float:
    """Algorithmic HBM bytes of the convolution launches of one forward pass over `batch` crops (what `roofline.traffic` is compared
    with): per launch its input + output (+ residual) activations at 2 bytes per element and its weights once, with the expanded tensor
    of the one-launch Fused-MBConv blocks and of the fused 8 x 8 fronts counted as what those launches really move (tools/layer_breakdown.py
    walks the same table per layer; SE gates: 4 bytes per gated channel and sample)."""
    B, t = float(batch), 0.0
    for b in effnetv2.blocks():
        o, i2 = b.out_hw * b.out_hw, b.in_hw * b.in_hw
        r = 2 if b.residual else 1
        if b.kind == "fused":
            if b.cexp == b.cin:
                t += 2.0 * (B * i2 * b.cin + B * o * b.cout * r + 9 * b.cin * b.cout)
            elif b.cexp <= 256 and b.cout <= 128:
                t += 2.0 * (B * i2 * b.cin + B * o * b.cout * r + 9 * b.cin * b.cexp + b.cexp * b.cout)
            else:
                t += 2.0 * (B * i2 * b.cin + B * o * b.cexp + 9 * b.cin * b.cexp)
                t += 2.0 * (B * o * b.cexp + B * o * b.cout * r + b.cexp * b.cout)
        else:
            t += 2.0 * (B * i2 * (b.cin + b.cexp) + b.cin * b.cexp)
            t += 2.0 * (B * o * b.cexp + B * o * b.cout * r + b.cexp * b.cout) + 4.0 * B * b.cexp
    return t + 2.0 * (B * 64 * 640 + 640 * 1280) + 4.0 * B * 64 * 1280


# algorithmic FLOPs per unit: one frame through the pose stage (backbone 15.966 + head 0.024 GMAC), one window through the
# match stage (MLP + tuple projections + attention + discriminator, support K / V amortised; per class 2 T^2 128 MACs of attention)
POSE_FLOPS_PER_FRAME = 2.0 * 15.99e9


def ar_flops_per_window(way: int, L: int = 30, J: int = 122) -> float:
    T = L * (L - 1) // 2
    return 2.0 * (L * (3 * J * 6 * J + 6 * J * 256) + 2 * T * 512 * 128 + way * 2 * T * T * 128 + T * 128 * L + T * L * 256 + 256 * 64 + 64)      # (query K / V projections as the reference writes them)


def ar_proto_algorithmic_bytes(windows: int, way: int, L: int = 30) -> float:
    """Algorithmic HBM bytes of one all-classes attention launch: per window its K fragment image (16-bit), the column statistics of every
    class (f32) and its V image (f32); the support set's K / V^T images once."""
    T = L * (L - 1) // 2
    Tp = (T + 31) // 32 * 32
    return windows * (Tp * 128 * 2 + way * Tp * 4 + Tp * 128 * 4) + 2.0 * way * Tp * 128 * 2


def dw_macs_per_crop() -> int:
    """MACs of the 57 depthwise 3x3 convolutions per crop"""
    return sum(9 * b.cexp * b.out_hw * b.out_hw for b in effnetv2.blocks() if b.kind != "fused")


class _HpeBase:
    L, J = 30, 122
    precision = "f16"          # 16-bit storage type of the pose backbone (isb_hpe_cfg.precision 0 / 2) and of the AR attention operands

    def _setup_hpe(self, args, rank, dev):
        import torch
        self.torch = torch
        self.B = args.batch or 256
        self.dev = dev
        self.bb_state = effnetv2.make_state(0)
        # Steps in flight. One step = one 256-frame batch through the pose stage. The library's own arrangement splits a batch into two
        # 128-frame halves on two streams; keeping SEVERAL whole batches in flight instead -- consecutive steps on one-lane engines, each
        # on its own stream, started group by group -- runs 256-frame launches (better tiles, fewer workgroup rounds) that still fill each
        # other's gaps: 15.05 vs 15.7 ms per 256 frames (tools/exp_lane_offset.py 8 256). Consecutive batches are independent (the pose
        # ring that carries over is updated in step order on the match stream), every step still does all of its work, and the
        # timed region ends on a device-wide synchronize. ISB_BENCH_INFLIGHT=1 = one batch at a time on the library's two lanes.
        # three steps need a hardware queue per stream: HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues, and three pose
        # streams + the match stream + the caller's on four queues serialise (19.3 ms per step against 17.55 on eight queues; two steps in
        # flight: 17.75; four: 18.1-20.9). The LIBRARY asks for eight queues when it is loaded (isb_hw_queues, include/isbfsar.h) unless the
        # caller's environment holds a value; bench.py loads it before anything touches the GPU.
        from isbfsar_amd import _lib
        self.hw_queues = int(_lib.lib().isb_hw_queues(None))     # what the HIP runtime reads (the library asks for 8 when it is loaded)
        dflt = "3" if self.hw_queues >= 8 else "2"
        self.n_flight = max(1, min(4, int(os.environ.get("ISB_BENCH_INFLIGHT", dflt)))) if self.B >= 64 else 1
        self.hpe_precision_arg = getattr(args, "hpe_precision", "f16")
        # ONE copy of the weights: the engines of the steps in flight are children of the first (isb_hpe_create_shared) -- they read
        # its model and own their streams and workspaces. ISB_BENCH_SHARED=0: an engine with its own weights per step (round 5).
        self.hpe = self._make_hpe(self.hpe_precision_arg)
        self.hpes = [self.hpe] + [self._make_hpe(self.hpe_precision_arg, parent=self.hpe) for _ in range(self.n_flight - 1)]
        self.pose_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_flight)] if self.n_flight > 1 else None
        self.step_no = 0
        self.lat_ev = None            # armed by in_flight_report(): [start, end] event pairs of the steps of a timed loop
        self.serial = False           # the roofline pass: one engine, whole-batch launches one after the other on the current stream
        self.pose_done = None
        self.precision = self.hpe.precision if self.hpe.precision != "bf16_f16tail" else "bf16"
        self.frames_host = synth.frames(self.B, seed=10_000 * rank)
        self.bbox_host = synth.bboxes(self.B, seed=10_000 * rank)
        self.frames = torch.from_numpy(self.frames_host).cuda(dev)
        self.bbox = torch.from_numpy(self.bbox_host).cuda(dev)

    def _make_hpe(self, precision, parent=None):
        if parent is not None and os.environ.get("ISB_BENCH_SHARED", "1") != "0":
            return parent.share()                                    # (the parent's lane count, taken at this moment, included)
        e = HpeEngine(device=self.dev, max_batch=min(self.B, int(os.environ.get("ISB_HPE_MICROBATCH", "1024"))), precision=precision)
        e.load_weights(self.bb_state)
        e.set_joint_map(np.load(os.path.join(_ASSETS, "32_to_122.npy")), None)   # skeleton=None -> 122 joints
        if self.n_flight > 1:
            e.set_lanes(1)
        return e

    def _pose(self, frames, bbox):
        """The pose stage of one step -> (joints, valid). One batch in flight: on the current stream. Two: on the step's own engine
        and stream, behind the inputs the current stream has produced so far; `self.pose_done` is the event a consumer stream waits
        for. Pairs of steps start together."""
        torch = self.torch
        if self.pose_streams is None or self.serial:
            self.pose_done = None
            return self.hpe.forward(frames, bbox)
        k = self.step_no % self.n_flight
        self.step_no += 1
        ps = self.pose_streams[k]
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        ps.wait_event(ready)
        frames.record_stream(ps)                              # (the caching allocator must not recycle them under the pose stream)
        bbox.record_stream(ps)
        with torch.cuda.stream(ps):
            if self.lat_ev is not None:                       # step latency under overlap: from the step's first launch ...
                self.lat_ev.append([torch.cuda.Event(enable_timing=True), None])
                self.lat_ev[-1][0].record(ps)
            joints, valid = self.hpes[k].forward(frames, bbox)
            self.pose_done = torch.cuda.Event(enable_timing=self.lat_ev is not None)
            self.pose_done.record(ps)
            if self.lat_ev is not None:
                self.lat_ev[-1][1] = self.pose_done           # ... to the end of its pose stage (the pipeline moves this to the end of its match stage)
        if k == self.n_flight - 1:                            # a group is enqueued: the next group starts when all of it is through
            for a in self.pose_streams:
                for b in self.pose_streams:
                    if a is not b:
                        a.wait_stream(b)
        return joints, valid

    def in_flight_report(self, steps):
        """First-class fields of the JSON line: with K steps in flight `value` / `ms_per_step` are OVERLAPPED throughput --
        a step is finished every ms_per_step, it is not that long. Reported beside them, measured in the same run:
          one_step_in_flight  -- the same steps one at a time through ONE engine (the batch split into the library's two half-batch lanes,
                                 everything on the caller's stream): what a single isb_hpe_forward call per step gives, and a step's latency
                                 when nothing else runs;
          step_latency_ms     -- first launch of a step -> end of its last stage while K steps are in flight (median / max over the steps)."""
        torch = self.torch
        if self.n_flight <= 1 or self.pose_streams is None or getattr(self, "host_input", False):
            return None
        steps = max(3, min(steps, 10))
        torch.cuda.synchronize()
        for _ in range(2):
            self.step()
        self.lat_ev = []
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        lat = sorted(a.elapsed_time(b) for a, b in self.lat_ev)
        self.lat_ev = None
        self.hpe.set_lanes(2)
        self.serial = True
        for _ in range(2):
            self.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        self.serial = False
        self.hpe.set_lanes(1)
        m, w, n = self.hpe.memory()
        return {"steps_in_flight": self.n_flight, "hw_queues": self.hw_queues,
                "one_step_in_flight": {"value": round(self.units_per_step() / dt, 3), "ms_per_step": round(dt * 1e3, 4)},
                "step_latency_ms": {"p50": round(lat[len(lat) // 2], 3), "max": round(lat[-1], 3), "steps": len(lat)},
                "engines": {"on_one_model": n, "model_bytes": m, "workspace_bytes_per_engine": w}}

    def _hpe_roofline(self, steps):
        self.torch.cuda.synchronize()
        self.serial = True
        self.hpe.profile(True)
        for _ in range(steps):
            self.step()
        self.torch.cuda.synchronize()
        ms, launches = self.hpe.profile_read()
        ms_dw, launches_dw = self.hpe.profile_read_dw()
        self.hpe.profile(False)
        self.serial = False
        flops = 2.0 * igemm_macs_per_crop() * self.B * steps
        achieved = flops / (ms / 1e3) / 1e12
        traffic = None
        tj = latest_traffic_json()
        if tj:                      # HBM bytes of the conv_igemm launches of one forward pass, from the PMC passes
            with open(tj) as f:     # (FETCH_SIZE / WRITE_SIZE, collected separately; see profiles/README.md)
                t = json.load(f)["hpe_b256"]["conv_igemm"]["hbm_bytes_per_forward"]
            traffic = t * self.B / 256.0
        alg = igemm_algorithmic_bytes(self.B)
        return {"bound": "mfma", "kernel": "conv_igemm / gemm1x1 family (all convolution launches of a forward pass)",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS_BF16, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS_BF16, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per forward pass over all convolution launches (PMC, measured at B=256)",
                "algorithmic_bytes": alg, "traffic_over_algorithmic": round(traffic / alg, 3) if traffic else None,
                "avg_launch_ms": round(ms / max(launches, 1), 5), "launches": int(launches),
                "flops_per_step": flops / steps,
                "power": latest_power("hpe", achieved, MFMA_PEAK_TFLOPS_BF16),
                # the fused MBConv fronts (mbfront8 / mbfront16) are family launches that carry their blocks' depthwise + SiLU + pool work:
                # every fusion of that kind moves time INTO the family and takes a depthwise launch away. Comparable across rounds:
                "with_depthwise": {
                    "what": "all convolution AND depthwise launches of a forward pass: their algorithmic FLOPs / their summed HIP-event time",
                    "achieved": round((flops + 2.0 * dw_macs_per_crop() * self.B * steps) / ((ms + ms_dw) / 1e3) / 1e12, 2),
                    "frac": round((flops + 2.0 * dw_macs_per_crop() * self.B * steps) / ((ms + ms_dw) / 1e3) / 1e12 / MFMA_PEAK_TFLOPS_BF16, 4),
                    "family_ms_per_pass": round(ms / steps, 3), "depthwise_ms_per_pass": round(ms_dw / steps, 3),
                    "depthwise_launches_per_pass": int(launches_dw // max(steps, 1))}}

    def step_flops(self):
        """algorithmic FLOPs of one step of this workload (for roofline.step_frac)"""
        return POSE_FLOPS_PER_FRAME * self.B

    def _cpu_hpe(self, n, iters):
        """The fp32 pose oracle on the first n frames of this rank's batch: (median seconds per frame, total seconds,
        parity of the GPU's poses for the same frames against the poses the timed oracle produced)."""
        from oracle import hpe_oracle as ho
        from oracle.effnetv2_oracle import EffNetV2LOracle
        import torch
        torch.set_num_threads(usable_cores())
        K = ho.intrinsics_matrix(384.025146484375, 384.025146484375, 319.09661865234375, 237.75723266601562)
        net = EffNetV2LOracle(self.bb_state, "f32")
        W = np.load(os.path.join(_ASSETS, "32_to_122.npy"))
        n = min(n, len(self.frames_host))
        poses = [None] * n

        def one_pass():
            for j in range(n):
                nk, r, H = ho.crop_params(self.bbox_host[j], K)
                crop = ho.warp(self.frames_host[j], H[0])
                lg = net.head(net.backbone(crop[None]))
                poses[j] = ho.postprocess(lg, nk, r, W, None)

        med, total = median_time(one_pass, warm=1, iters=iters)
        joints, valid = self.hpe.forward(self.frames_host[:n], self.bbox_host[:n])
        ok = [j for j in range(n) if poses[j] is not None and valid[j]]
        par = {"n_frames": n, "valid_agree": bool(all((poses[j] is not None) == bool(valid[j]) for j in range(n)))}
        if ok:
            ref = np.stack([poses[j] for j in ok])
            g = joints[ok].astype(np.float64)
            # root-centred pose = what the AR stage consumes (main.py:103); absolute = what estimate() returns
            par["joints_rootcentred_maxabs"] = float(np.abs((g - g[:, :1]) - (ref - ref[:, :1])).max())
            par["joints_maxabs"] = float(np.abs(g - ref).max())
            par["joints_l2_per_frame"] = float(np.sqrt(((g - ref) ** 2).sum(axis=(1, 2))).max())
        return med / n, total, par


class HpeWorkload(_HpeBase):
    name = "hpe"
    metric = "frames/sec (640x480 frame -> crop -> EfficientNetV2-L -> 3D pose, 122 joints)"
    unit = "frames/s"

    def __init__(self, args, rank, world, dev):
        self._setup_hpe(args, rank, dev)
        self.world = world
        self.host_input = bool(getattr(args, "host_input", False))
        self.pipelined = bool(getattr(args, "pipelined", False)) and self.host_input
        if self.host_input:          # pinned host frames: what a capture thread would hand over (utils/input.py -> main.py:74)
            self.frames_pinned = self.torch.from_numpy(self.frames_host).pin_memory().numpy()
        if self.pipelined:           # two capture buffers, two batches in flight: submit batch k + 1, then wait for batch k
            self.frames_pinned2 = self.torch.from_numpy(self.frames_host).pin_memory().numpy()
            self.k = 0
            self.inflight = 0

    def units_per_step(self):
        return self.B

    def step(self):
        if self.pipelined:           # steady state: one submit + one wait per step
            self.hpe.submit((self.frames_pinned, self.frames_pinned2)[self.k & 1], self.bbox_host)
            self.k += 1
            self.inflight += 1
            if self.inflight == 2:
                self.out = self.hpe.wait()
                self.inflight -= 1
        elif self.host_input:        # H2D of the frames + kernels + D2H of the poses + synchronise, inside the step
            self.out = self.hpe.forward(self.frames_pinned, self.bbox_host)
        else:
            self.out = self._pose(self.frames, self.bbox)

    def roofline(self, steps):
        return self._hpe_roofline(steps)

    def cpu_baseline(self, sample, iters=10):
        n = sample or 8                  # HPE batch 8; ~0.8 s per pass on the box's 16 host cores
        spf, total, self.parity = self._cpu_hpe(n, iters)
        return {"value": round(1.0 / spf, 3), "unit": "frames/s", "cores": usable_cores(), "kind": "port",
                "sample": f"median of {iters} passes over {n} frames through the fp32 CPU oracle (numpy geometry + torch-CPU "
                          f"EfficientNetV2-L), {total:.1f} s of CPU work"}

    def config(self, world):
        return {"workload": f"BASELINE configs[1]: B={self.B} synthetic 640x480 frames/GPU, HPE only "
                            "(homography crop, EfficientNetV2-L bf16, head, decode, reconstruction)"
                            + (" -- frames in pinned HOST memory, H2D + D2H inside the step (isb_hpe_forward_host)" if self.host_input else "")
                            + ("; two batches in flight (isb_hpe_submit_host / isb_hpe_wait_host)" if self.pipelined else ""),
                "input": ("host (pinned), two batches in flight" if self.pipelined else "host (pinned)") if self.host_input else "resident in HBM",
                "per_gpu_batch": self.B, "n_joints": self.J, "parallelism": f"dp{world}",
                "steps_in_flight": 1 if (self.host_input or self.pose_streams is None) else self.n_flight}


class PipelineWorkload(_HpeBase):
    name = "pipeline"
    metric = "end-to-end pipelines/sec (frame -> 3D pose -> 30-frame window embed -> 60-class match + open-set score)"
    unit = "pipelines/s"
    N_CAM = 8

    def __init__(self, args, rank, world, dev):
        import torch
        self._setup_hpe(args, rank, dev)
        self.world = world
        self.way = args.way
        self.ar_precision = args.precision
        if self.B % self.N_CAM:
            raise SystemExit(f"--batch must be a multiple of {self.N_CAM} cameras")
        self.steps_per_cam = self.B // self.N_CAM
        self.ar_state = weights.make_ar_state(self.L, self.J, seed=1)
        self.ss = synth.skeleton_windows(self.way, self.L, self.J, seed=101)
        self.ar = ArEngine(self.L, self.J, self.way, device=dev, precision=self.ar_precision, max_batch=self.B)
        self.ar.load_weights(self.ar_state)
        self.ar.set_support(poses=self.ss)
        # per-camera ring: L-1 poses of history + this step's poses
        hist = synth.skeleton_windows(self.N_CAM, self.L - 1, self.J, seed=555 + rank).reshape(self.N_CAM, self.L - 1, self.J, 3)
        self.ring = torch.zeros((self.N_CAM, self.L - 1 + self.steps_per_cam, self.J, 3), dtype=torch.float32, device=f"cuda:{dev}")
        self.ring[:, : self.L - 1] = torch.from_numpy(hist).cuda(dev)
        self.checked = False
        # N > 1: the match stage of step i and the all-gather of its records run on a second stream, so the collective
        # travels over xGMI while the pose stage of step i+1 computes (consecutive batches are independent; the pose ring
        # that carries over lives on the first stream). On one GPU the same arrangement changes nothing (measured:
        # the two pose lanes already fill the chip), so N = 1 runs the stages back to back.
        self.side = (torch.cuda.Stream(device=dev)
                     if (world > 1 and os.environ.get("ISB_BENCH_OVERLAP", "1") != "0") or self.n_flight > 1 else None)
        self.gather, self.force = make_gather(args, world, dev)

    def units_per_step(self):
        return self.B

    def step(self):
        torch = self.torch
        bbox = self.bbox
        if getattr(self, "det", None) is not None:
            # the reference's whole per-frame path (hpe.py:51-79): detector -> most confident person box -> pose. The synthetic
            # detector weights find no person in synthetic frames, so a frame without a box keeps its synthetic one: every stage
            # does its full work on every frame
            boxes, confs = self.det.forward(self.frames)
            det_bbox, found = self.hpe.select_person(boxes, confs, 0.3)
            bbox = torch.where(found.bool()[:, None], det_bbox, self.bbox)
        joints, valid = self._pose(self.frames, bbox)                            # [B,122,3]
        # everything behind the pose stage -- the pose ring (carried from step to step), the windows, the match stage, the all-gather
        # -- runs in step order on the match stream; with one batch in flight on one GPU that is the current stream
        ms = self.side if (not self.serial or self.world > 1) else None      # (one step at a time on one GPU: everything on the caller's stream)
        if ms is not None:
            ms.wait_stream(torch.cuda.current_stream())
            if self.pose_done is not None:
                ms.wait_event(self.pose_done)
            joints.record_stream(ms)
            valid.record_stream(ms)
        with (torch.cuda.stream(ms) if ms is not None else contextlib.nullcontext()):
            if not self.checked:
                assert bool(valid.all().item()), "synthetic frames are expected to give in-FOV poses"
                self.checked = True
            self.ring[:, self.L - 1:] = joints.view(self.N_CAM, self.steps_per_cam, self.J, 3)
            windows = pose_windows(self.ring, self.L)                             # [B,L,3J], root-centred
            self.ring[:, : self.L - 1] = self.ring[:, self.steps_per_cam:].clone()  # slide the history
            self._match(windows)
            if self.lat_ev:                                   # (armed, and this step went through _pose's in-flight branch)
                self.lat_ev[-1][1] = torch.cuda.Event(enable_timing=True)
                self.lat_ev[-1][1].record(torch.cuda.current_stream())

    def _match(self, windows):
        logits, is_true, embed = self.ar.infer(windows, want_embed=self.world > 1)
        if self.world > 1:
            # ONE all-gather of the packed per-window records over RCCL/xGMI
            self.out = all_gather_records(pack_records(logits, is_true, embed), force=self.force, gather=self.gather)
        else:
            self.out = (logits, is_true)

    def roofline(self, steps):
        r = self._hpe_roofline(steps)
        # the match stage's dominant kernel in the same line: ar_proto_all over this step's windows (HIP events on its launch stream)
        self.serial = True              # (alone on the chip, like the convolution family above: a kernel's duration, not its share of an overlap)
        self.ar.profile(True)
        for _ in range(steps):
            self.step()
        self.torch.cuda.synchronize()
        ms, launches = self.ar.profile_read()
        self.ar.profile(False)
        self.serial = False
        T = self.L * (self.L - 1) // 2
        fl = self.B * self.way * 2 * (2 * T * T * 128)
        ach = fl / (ms / max(launches, 1) / 1e3) / 1e12 if ms > 0 else None
        traffic = None
        tj = latest_traffic_json()
        if tj:
            with open(tj) as f:
                traffic = json.load(f).get("ar_b1024", {}).get("ar_proto", {}).get("hbm_bytes_per_launch")
        r["ar_proto_all"] = {"achieved": round(ach, 2) if ach else None, "peak": MFMA_PEAK_TFLOPS_BF16, "unit": "TFLOP/s",
                             "frac": round(ach / MFMA_PEAK_TFLOPS_BF16, 4) if ach else None,
                             "avg_launch_ms": round(ms / max(launches, 1), 4), "launches": int(launches), "windows_per_launch": self.B,
                             "traffic": traffic, "traffic_unit": "HBM bytes of ONE 1024-window launch, way 60 (PMC, `--workload ar`)",
                             "algorithmic_bytes": ar_proto_algorithmic_bytes(1024, 60, self.L),
                             "traffic_over_algorithmic": round(traffic / ar_proto_algorithmic_bytes(1024, 60, self.L), 2) if traffic else None}
        r["power"] = latest_power("pipeline", r.get("achieved"), MFMA_PEAK_TFLOPS_BF16) or r.get("power")
        return r

    def step_flops(self):
        return (POSE_FLOPS_PER_FRAME + ar_flops_per_window(self.way, self.L, self.J)) * self.B

    def _initial_ring(self, rank):
        hist = synth.skeleton_windows(self.N_CAM, self.L - 1, self.J, seed=555 + rank).reshape(self.N_CAM, self.L - 1, self.J, 3)
        ring = self.torch.zeros((self.N_CAM, self.L - 1 + self.steps_per_cam, self.J, 3), dtype=self.torch.float32, device=f"cuda:{self.dev}")
        ring[:, : self.L - 1] = self.torch.from_numpy(hist).cuda(self.dev)
        return ring

    def verify_gather(self, rank, world):
        """N > 1 (every rank calls this: one collective step): the all-gathered records of one step from a known state must be,
        BIT FOR BIT, what ONE process computes for every rank's frames without any collective (SURVEY.md 8e correctness check).
        Rank 0 recomputes the other ranks' shards locally (their inputs are seeded by rank) and compares."""
        torch = self.torch
        self.ring = self._initial_ring(rank)
        self.step()
        torch.cuda.synchronize()
        got = self.out
        if rank != 0 or not torch.is_tensor(got):
            return None
        ref = []
        for q in range(world):
            frames = torch.from_numpy(synth.frames(self.B, seed=10_000 * q)).cuda(self.dev)
            bbox = torch.from_numpy(synth.bboxes(self.B, seed=10_000 * q)).cuda(self.dev)
            ring = self._initial_ring(q)
            joints, _ = self.hpe.forward(frames, bbox)
            ring[:, self.L - 1:] = joints.view(self.N_CAM, self.steps_per_cam, self.J, 3)
            logits, is_true, embed = self.ar.infer(pose_windows(ring, self.L), want_embed=True)
            ref.append(pack_records(logits, is_true, embed))
        ref = torch.cat(ref, dim=0)
        torch.cuda.synchronize()
        same = bool(got.shape == ref.shape and torch.equal(got, ref))
        return {"gathered_records": list(got.shape), "bit_equal_to_unsharded": same,
                "how": "one step from the initial pose rings; rank 0 recomputed every rank's shard (inputs seeded by rank) without a collective"}

    def _timed(self, steps, warm=2):
        torch = self.torch
        for _ in range(warm):
            self.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def extras(self, args):
        """Measured in the same run as the headline, reported inside `config` (N = 1 only):
          * value_with_detector -- the reference's whole per-frame path: the YOLOv4 person detector and the box selection in
            front of the pose stage (BASELINE's configs hand the boxes in);
          * value_ar_bf16 -- the pipeline with the AR attention on bf16 operands (round 3's default; fp16 operands -- 11 significant
            bits at the same matrix rate, logits 2-13x closer to the fp32 reference, tests/test_ar_gpu.py -- are the default now);
          * value_bf16x3 -- the pipeline with the AR attention at its fp32-grade precision (hi + lo split, 3 MFMAs per
            product; the reference's TRXOS is fp32);
          * whole_batch_2048 -- BASELINE configs[3]'s WHOLE batch (2048 frames -> 2048 windows) on this one GPU."""
        out = {}
        if self.B <= 256:
            from isbfsar_amd import yolov4
            from isbfsar_amd.det_engine import DetEngine
            self.det = DetEngine(device=self.dev, max_batch=self.B)
            self.det.load_weights(yolov4.make_state(0))
            dt = self._timed(max(3, min(args.steps, 10)))
            out["value_with_detector"] = round(self.B / dt, 3)
            out["ms_per_step_with_detector"] = round(dt * 1e3, 4)
            self.det.close()
            self.det = None
        if self.ar_precision == "f16":
            ar0 = self.ar
            self.ar = ArEngine(self.L, self.J, self.way, device=self.dev, precision="bf16", max_batch=self.B)
            self.ar.load_weights(self.ar_state)
            self.ar.set_support(poses=self.ss)
            dt = self._timed(max(3, min(args.steps, 10)))
            out["value_ar_bf16"] = round(self.B / dt, 3)
            out["ms_per_step_ar_bf16"] = round(dt * 1e3, 4)
            self.ar.close() if hasattr(self.ar, "close") else None
            self.ar = ar0
        if self.ar_precision != "bf16x3":
            ar0 = self.ar
            self.ar = ArEngine(self.L, self.J, self.way, device=self.dev, precision="bf16x3", max_batch=self.B)
            self.ar.load_weights(self.ar_state)
            self.ar.set_support(poses=self.ss)
            dt = self._timed(max(3, min(args.steps, 10)))
            out["value_bf16x3"] = round(self.B / dt, 3)
            out["ms_per_step_bf16x3"] = round(dt * 1e3, 4)
            self.ar.close() if hasattr(self.ar, "close") else None
            self.ar = ar0
        if self.B <= 256:
            out.update(self.hpe_precision_report())
        if self.B == 256 and not args.batch:
            import copy
            a2 = copy.copy(args)
            a2.batch = 2048
            big = PipelineWorkload(a2, 0, 1, self.dev)
            dt = big._timed(5, warm=3)
            out["whole_batch_2048"] = {"value": round(2048 / dt, 3), "unit": self.unit, "ms_per_step": round(dt * 1e3, 4), "steps": 5,
                                       "note": "configs[3]'s whole batch on one GPU: 8 cameras x 256 steps -> 2048 windows, pose "
                                               "micro-batches of up to 1024 frames"}
            del big
            self.torch.cuda.empty_cache()
        return out

    def hpe_precision_report(self, n=32):
        """The pose backbone's 16-bit storage layouts side by side: the pipeline's rate with each
        (value_hpe_<precision>; the headline runs the default, fp16 everywhere) and every layout's distance to the fp32 CPU
        definition on BOTH synthetic weight profiles -- "default" (near-constant features: benign) and "signal" (activations that
        carry the input, peaked heat-maps: the regime of a trained MetrABS) -- for the first n frames of the batch."""
        from oracle import hpe_oracle as ho
        from oracle.effnetv2_oracle import EffNetV2LOracle
        torch = self.torch
        torch.set_num_threads(usable_cores())
        K = ho.intrinsics_matrix(384.025146484375, 384.025146484375, 319.09661865234375, 237.75723266601562)
        W = np.load(os.path.join(_ASSETS, "32_to_122.npy"))
        fr, bb = self.frames_host[:n], self.bbox_host[:n]
        crops = np.stack([ho.warp(fr[j], ho.crop_params(bb[j], K)[2][0]) for j in range(n)])
        out, parity = {}, {}
        states = {"default": self.bb_state, "signal": effnetv2.make_state(0, "signal", head_gain=0.5)}
        refs = {}
        for prof, state in states.items():
            net = EffNetV2LOracle(state, "f32")
            lg = np.concatenate([net.head(net.backbone(crops[i:i + 16])) for i in range(0, n, 16)])
            p2, p3 = ho.decode(lg)
            poses = [ho.postprocess(lg[j:j + 1], *ho.crop_params(bb[j], K)[:2], W, None) for j in range(n)]
            refs[prof] = (p3, poses)
        hpe0 = self.hpe
        for prec in ("f16", "bf16_f16tail", "bf16"):
            if prec == hpe0.precision:
                e = hpe0
            else:
                e = HpeEngine(device=self.dev, max_batch=hpe0.max_batch, precision=prec)
                e.set_joint_map(W, None)
            for prof, state in states.items():
                if not (e is hpe0 and prof == "default"):
                    e.load_weights(state)
                joints, valid = e.forward(fr, bb)
                _, lg = e.backbone(crops)
                p3_ref, poses = refs[prof]
                ok = [j for j in range(n) if poses[j] is not None and valid[j]]
                g = joints[ok].astype(np.float64)
                ref = np.stack([poses[j] for j in ok])
                per_frame = np.sort(np.abs(g - ref).reshape(len(ok), -1).max(axis=1))
                parity.setdefault(prof, {})[prec] = {
                    "decoded3d_maxabs": float(np.abs(ho.decode(lg)[1] - p3_ref).max()),
                    "joints_maxabs": float(per_frame[-1]),
                    # the absolute pose goes through a least-squares fit that amplifies heat-map noise: the distribution over the
                    # frames, not only its maximum
                    "joints_p50": float(np.median(per_frame)),
                    "joints_p99": float(per_frame[min(len(ok) - 1, int(np.ceil(0.99 * len(ok))) - 1)]),
                    "joints_rootcentred_maxabs": float(np.abs((g - g[:, :1]) - (ref - ref[:, :1])).max()), "n_frames": len(ok)}
            e.load_weights(self.bb_state)
            if e is not hpe0:
                hpes0 = self.hpes                       # the pipeline at this layout: as many engines as steps are in flight
                if self.n_flight > 1:
                    e.set_lanes(1)
                self.hpes = [e] + [self._make_hpe(prec, parent=e) for _ in range(self.n_flight - 1)]
                self.hpe = e
                dt = self._timed(5)
                out[f"value_hpe_{prec}"] = round(self.B / dt, 3)
                out[f"ms_per_step_hpe_{prec}"] = round(dt * 1e3, 4)
                for x in self.hpes:
                    x.close()
                self.hpes, self.hpe = hpes0, hpe0
        out["parity_hpe_precisions_vs_fp32_oracle"] = parity
        return out

    def cpu_baseline(self, sample, iters=10):
        from oracle.ar_oracle import TRXOSOracle
        n = sample or 8                  # per pass: 8 frames (~0.8 s) and 8 windows (~0.7 s) on 16 host cores
        spf, t_hpe, par = self._cpu_hpe(n, iters)
        net = TRXOSOracle(self.ar_state, self.L, self.J)
        sf = net.mlp(self.ss)
        q = synth.skeleton_windows(n, self.L, self.J, seed=9)
        ref = {}
        med, t_ar = median_time(lambda: ref.update(net.forward(None, self.way, q, ss_features=sf)), warm=1, iters=iters)
        spw = med / len(q)
        logits, is_true, embed = self.ar.infer(q, want_embed=True)
        par.update(ar_parity(logits, is_true, embed, ref))
        self.parity = par
        return {"value": round(1.0 / (spf + spw), 3), "unit": "pipelines/s", "cores": usable_cores(), "kind": "port",
                "sample": f"median of {iters} passes: {n} frames through the fp32 pose oracle ({spf * 1e3:.0f} ms/frame) + "
                          f"{len(q)} windows through the AR oracle ({spw * 1e3:.0f} ms/window); pipelines/s = 1/(sum); "
                          f"{t_hpe + t_ar:.1f} s of CPU work"}

    def config(self, world):
        return {"workload": f"BASELINE configs[3] per-GPU shard: {self.B} synthetic 640x480 frames/GPU "
                            f"({self.N_CAM} cameras x {self.steps_per_cam} steps) -> HPE (EfficientNetV2-L, 122 joints) -> "
                            f"30-frame windows -> AR (way={self.way}) -> open-set score",
                "per_gpu_batch": self.B, "seq_len": self.L, "n_joints": self.J, "way": self.way,
                # consecutive steps (independent batches) overlap: each step's pose stage is ONE lane of whole-batch launches on its
                # own engine + stream, groups of steps start together, the pose ring / windows / match stage / all-gather follow in step
                # order on a third stream. 1 = one step at a time, its batch split into two half-batch lanes (ISB_BENCH_INFLIGHT=1)
                "steps_in_flight": self.n_flight,
                "ar_precision": self.ar_precision,
                "hpe_precision": self.hpe.precision + {"f16": " (IEEE fp16 weights and activations in every stage -- the reference's TensorRT precision, "
                                                              "7_create_engines.py:10; f32 accumulate, f32 head / decode, f64 reconstruction)",
                                                       "bf16_f16tail": " (bf16 storage; fp16 in the two 8x8 stages and the 640->1280 conv)"}.get(self.hpe.precision, ""),
                "parallelism": f"dp{world} (frames sharded; one all-gather of per-window records"
                               + (": RCCL behind the C ABI (isb_dist_all_gather)" if self.gather is not None else "")
                               + (", on a second stream beside the next step's pose stage)" if self.side is not None else ")")}


class StreamWorkload(_HpeBase):
    """BASELINE configs[4]: one camera feed per GPU, per-frame step = HPE on 1 frame + AR on the current
    30-frame sliding window against a 120-class support set, the whole step captured in ONE hipGraph
    (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream the library launches on)."""
    name = "stream"
    metric = "sustained per-feed steps/sec (1 frame -> pose -> sliding 30-frame window -> 120-class match + open-set), hipGraph replay"
    unit = "steps/s"

    def __init__(self, args, rank, world, dev):
        import torch
        args.batch = 1
        self._setup_hpe(args, rank, dev)
        self.world = world
        self.way = 120 if args.way == 60 else args.way
        self.ar_precision = args.precision
        self.ar_state = weights.make_ar_state(self.L, self.J, seed=1)
        self.ss = synth.skeleton_windows(self.way, self.L, self.J, seed=101)
        self.ar = ArEngine(self.L, self.J, self.way, device=dev, precision=self.ar_precision, max_batch=1)
        self.ar.load_weights(self.ar_state)
        self.ar.set_support(poses=self.ss)
        hist = synth.skeleton_windows(1, self.L, self.J, seed=777 + rank).reshape(1, self.L, self.J, 3)
        self.ring = torch.from_numpy(hist).cuda(dev).contiguous()
        self.graph = None
        self.out = None
        # N > 1: every rank ends a step holding every feed's [probabilities-to-be | open-set score] record; the all-gather
        # (RCCL behind the C ABI) is part of the captured step
        self.gather, self.force = make_gather(args, world, dev)
        self.gathered = None
        if self.gather is not None:
            self.gathered = torch.empty((self.gather.world, self.way + 1), dtype=torch.float32, device=f"cuda:{dev}")
        # warm-up outside capture (workspaces are allocated on first use), then capture one step
        for _ in range(2):
            self._step_eager()
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                self._step_eager()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    self._step_eager()
            torch.cuda.synchronize()
            self.graph = g
        except Exception as e:          # report, do not hide: the JSON line says graph=false
            self.graph_error = repr(e)
            torch.cuda.synchronize()

    def _step_eager(self):
        joints, valid = self.hpe.forward(self.frames, self.bbox)                 # [1,122,3]
        self.ring.copy_(self.torch.cat([self.ring[:, 1:], joints.view(1, 1, self.J, 3)], dim=1))
        windows = pose_windows(self.ring, self.L)                                 # [1,L,3J]
        logits, is_true, _ = self.ar.infer(windows)
        if self.gather is not None:
            self.gather.all_gather_into(pack_records(logits, is_true), self.gathered)
            self.out = (self.gathered, valid)
        else:
            self.out = (logits, is_true, valid)

    def units_per_step(self):
        return 1

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step_eager()

    def roofline(self, steps):
        r = self._hpe_roofline(steps) if self.graph is None else None
        lat = []
        torch = self.torch
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.step()
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        lat = np.sort(np.array(lat))
        extra = {"step_latency_ms": {"p50": round(float(lat[len(lat) // 2]), 4), "p99": round(float(lat[int(len(lat) * 0.99)]), 4)},
                 "hipgraph": self.graph is not None}
        if r is None:
            # under graph replay the per-launch events of the profile hook are not available; the roofline of
            # this latency-bound B=1 step is quoted from the un-captured run
            self.graph, g = None, self.graph
            r = self._hpe_roofline(steps)
            self.graph = g
        r.update(extra)
        return r

    def cpu_baseline(self, sample, iters=10):
        return PipelineWorkload.cpu_baseline(self, sample or 4, iters)

    def config(self, world):
        return {"workload": "BASELINE configs[4]: 1 camera feed per GPU, per-frame step = HPE (1 frame, 122 joints) + AR on the "
                            f"sliding 30-frame window, way={self.way}, hipGraph-captured={self.graph is not None}",
                "per_gpu_batch": 1, "seq_len": self.L, "n_joints": self.J, "way": self.way,
                "parallelism": f"dp{world} (one feed per GPU"
                               + (", all-gather of the per-feed records inside the captured step: isb_dist_all_gather)" if self.gather is not None
                                  else ", no collective)")}


class DetWorkload:
    """SURVEY 8f row 1 (not a BASELINE config): the YOLOv4 person detector the reference runs per frame before the pose
    stage (hpe.py:51-73): frame -> area resize -> CSPDarknet53 + SPP + PANet + heads (22.8 GFLOP) -> boxes / confs -> the
    person box. Reported beside the BASELINE workloads; its roofline is the whole step against the bf16 MFMA peak."""
    name = "det"
    metric = "frames/sec (640x480 frame -> YOLOv4 boxes + class confidences -> person box)"
    unit = "frames/s"
    precision = "bf16"

    def __init__(self, args, rank, world, dev):
        import torch
        from isbfsar_amd import yolov4
        from isbfsar_amd.det_engine import DetEngine
        self.torch = torch
        self.B = args.batch or 256                           # the frames of BASELINE configs[1] / [3]'s per-GPU shard
        self.state = yolov4.make_state(0)
        # one micro-batch (isb_det caps it at 256 frames): 110 launches whatever the batch, and at 64 frames most of them are too
        # small to fill the chip (14.0 k frames/s in 64-frame micro-batches, 18.4 k in 128, 20.1 k in 256)
        self.det = DetEngine(device=dev, max_batch=min(self.B, int(os.environ.get("ISB_DET_MICROBATCH", "256"))))
        self.det.load_weights(self.state)
        self.frames_host = synth.frames(self.B, seed=20_000 * (rank + 1))
        self.frames = torch.from_numpy(self.frames_host).cuda(dev)
        self.flops = 2.0 * yolov4.macs_per_frame()

    def units_per_step(self):
        return self.B

    def step(self):
        boxes, confs = self.det.forward(self.frames)
        self.out = (boxes, confs)

    def roofline(self, steps):
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            self.step()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / steps
        achieved = self.flops * self.B / (ms / 1e3) / 1e12
        return {"bound": "mfma", "kernel": "whole detector step (110 convolutions + pre-processing, SPP, concat, decode)",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS_BF16, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS_BF16, 4), "traffic": None, "avg_launch_ms": round(ms, 4), "launches": steps}

    def cpu_baseline(self, sample, iters=10):
        from oracle.yolov4_oracle import YoloV4Oracle, preprocess
        import torch
        torch.set_num_threads(usable_cores())
        n = sample or 4
        net = YoloV4Oracle(self.state, "f32")
        ref = {}

        def one_pass():
            img = np.stack([preprocess(f) for f in self.frames_host[:n]])
            ref["maps"] = net.raw_heads(img)
            ref["out"] = net.decode(ref["maps"])

        med, total = median_time(one_pass, warm=1, iters=iters)
        _, maps = self.det.debug(self.frames_host[:n]) if n <= self.det.max_batch else (None, None)
        if maps is not None:
            self.parity = {"n_frames": n, "maps_rel_l2": [round(float(np.linalg.norm(m - r) / np.linalg.norm(r)), 5) for m, r in zip(maps, ref["maps"])]}
        return {"value": round(n / med, 3), "unit": "frames/s", "cores": usable_cores(), "kind": "port",
                "sample": f"median of {iters} passes over {n} frames through the fp32 CPU definition (numpy pre-processing + torch-CPU "
                          f"YOLOv4 + numpy decode), {total:.1f} s of CPU work"}

    def config(self, world):
        return {"workload": f"SURVEY 8f row 1: B={self.B} synthetic 640x480 frames/GPU through the YOLOv4 person detector",
                "per_gpu_batch": self.B, "parallelism": f"dp{world}"}


def get(name: str):
    return {"hpe": HpeWorkload, "stream": StreamWorkload, "det": DetWorkload}.get(name, PipelineWorkload)
