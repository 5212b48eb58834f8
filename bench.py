#!/usr/bin/env python3
"""Benchmark of the hot path (frame -> pose -> embedding -> match) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--workload pipeline|hpe|ar]

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment when launched through
torch.distributed.run, backend "nccl" = RCCL). Started plainly with --gpus N > 1, this process does not touch the
GPU: it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD, forwards its output and
exits with its return code; a WORLD_SIZE that disagrees with --gpus is an error (exit 2), never a silent N=1 run.  A step is one pass
of the hot path over one batch of synthetic input already resident in HBM.  Units shard
across ranks with no data-path collective (weak scaling: per-GPU batch fixed); the only
collective is the all-gather of per-window results, inside the timed region for N>1.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# before the HIP runtime starts: one hardware queue per stream this process keeps busy (three pose streams, the match stream, the
# caller's stream, the library's copy stream); the runtime's default of four multiplexes them and serialises the third pose stream.
# The library asks for eight when it is loaded (isb_hw_queues, include/isbfsar.h: GPU_MAX_HW_QUEUES, unless the caller's environment
# holds a value), so load it before anything makes a HIP call.
from isbfsar_amd import _lib as _isb_lib  # noqa: E402

_isb_lib.lib()

import numpy as np  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0 / 3.0, "f32": 157.3}   # MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "pipeline", "hpe", "ar", "stream", "det"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = the BASELINE config's)")
    ap.add_argument("--way", type=int, default=60)
    ap.add_argument("--precision", default="f16", choices=["bf16", "f16", "bf16x3"],
                    help="operand type of the two tuple-attention contractions (f16: the default -- bf16's matrix rate, logits 2-13x closer to the fp32 reference)")
    ap.add_argument("--hpe-precision", default="f16", choices=["f16", "bf16", "bf16_f16tail"],
                    help="16-bit storage type of the pose backbone (isb_hpe_cfg.precision): f16 = IEEE fp16 in every stage, the default and "
                         "what the reference's TensorRT engines run; bf16; bf16_f16tail = round 3's mixed layout")
    ap.add_argument("--host-input", action="store_true",
                    help="hpe workload: frames start in (pinned) HOST memory each step, isb_hpe_forward_host copies them (PCIe-inclusive "
                         "rate: the reference's Runner pattern; never the headline value)")
    ap.add_argument("--pipelined", action="store_true",
                    help="with --host-input: two pinned capture buffers and two batches in flight (isb_hpe_submit_host / "
                         "isb_hpe_wait_host): batch k + 1's PCIe transfer hides behind batch k's kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="items per CPU-baseline iteration (0 = the workload's default)")
    ap.add_argument("--cpu-iters", type=int, default=10, help="timed CPU-baseline iterations (median reported; BASELINE.md 4)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra measurements the default pipeline line carries (fp32-grade AR rate, the whole 2048-frame batch)")
    ap.add_argument("--min-gpu-seconds", type=float, default=12.0,
                    help="keep stepping (untimed, after all measurements) until the GPU phase has lasted this long, so that a coarse "
                         "utilisation sampler sees it (N = 1 only; 0 = off)")
    return ap


def parse():
    return build_parser().parse_args()


def self_launch(args) -> int:
    """--gpus N > 1 without a torchrun environment: run the N ranks as a child torch.distributed.run job.
    Nothing in THIS process has touched the GPU (a process that has must never exec another program on this pool, and
    needs not: the child is an ordinary subprocess whose stdout -- the one JSON line of its rank 0 -- is ours)."""
    import socket
    import subprocess
    with socket.socket() as sk:                     # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["ISB_BENCH_CHILD"] = "1"
    print(f"bench.py: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------------------------
# AR workload: BASELINE.json configs[2] -- B windows of 30 x 122 joints vs 60 classes
# --------------------------------------------------------------------------------------------
class ArWorkload:
    name = "ar"
    L, J = 30, 122

    def __init__(self, args, rank, world, dev):
        import torch
        from isbfsar_amd import synth, weights
        from isbfsar_amd.engine import ArEngine

        self.torch = torch
        self.way = args.way
        self.B = args.batch or 1024
        self.precision = args.precision
        self.state = weights.make_ar_state(self.L, self.J, seed=1)
        self.ss = synth.skeleton_windows(self.way, self.L, self.J, seed=101)
        q = synth.skeleton_windows(self.B, self.L, self.J, seed=1000 + rank)
        self.q_host = q
        self.eng = ArEngine(self.L, self.J, self.way, device=dev, precision=self.precision, max_batch=min(self.B, 1024))
        self.eng.load_weights(self.state)
        self.eng.set_support(poses=self.ss)
        self.q = torch.from_numpy(q).cuda(dev)
        self.world = world
        self.out = None
        from bench_workloads import make_gather
        self.gather, self.force = make_gather(args, world, dev)

    def units_per_step(self):
        return self.B

    def step(self):
        logits, is_true, embed = self.eng.infer(self.q, want_embed=self.world > 1)
        if self.world > 1:
            from isbfsar_amd.dist import all_gather_records, pack_records
            # one fused all-gather of the packed per-window record
            self.out = all_gather_records(pack_records(logits, is_true, embed), force=self.force, gather=self.gather)
        else:
            self.out = (logits, is_true)

    # algorithmic work of the dominant kernel (ar_proto, all classes): S and P contractions
    def roofline(self, steps):
        T = self.L * (self.L - 1) // 2
        self.eng.profile(True)
        for _ in range(steps):
            self.step()
        self.torch.cuda.synchronize()
        ms, launches = self.eng.profile_read()
        self.eng.profile(False)
        chunk = min(self.B, 1024)
        flops_per_launch = chunk * self.way * 2 * (2 * T * T * 128)     # S^T and P^T MACs x2
        avg_s = ms / max(launches, 1) / 1e3
        achieved = flops_per_launch / avg_s / 1e12
        peak = MFMA_PEAK_TFLOPS[self.precision]
        traffic = None
        from bench_workloads import latest_traffic_json
        tj = latest_traffic_json()
        if tj and chunk == 1024 and self.way == 60 and self.precision in ("bf16", "f16"):
            with open(tj) as f:       # HBM bytes of one ar_proto launch (PMC passes, profiles/README.md)
                traffic = json.load(f).get("ar_b1024", {}).get("ar_proto", {}).get("hbm_bytes_per_launch")
        from bench_workloads import ar_proto_algorithmic_bytes
        alg = ar_proto_algorithmic_bytes(chunk, self.way, self.L)
        return {"bound": "mfma", "kernel": "ar_proto_all_kernel", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "algorithmic_bytes": alg, "traffic_over_algorithmic": round(traffic / alg, 2) if traffic else None,
                "avg_launch_ms": round(ms / max(launches, 1), 4), "launches": int(launches)}

    def step_flops(self):
        from bench_workloads import ar_flops_per_window
        return ar_flops_per_window(self.way, self.L, self.J) * self.B

    def cpu_baseline(self, sample, iters=10):
        from oracle.ar_oracle import TRXOSOracle
        import torch
        from bench_workloads import ar_parity, median_time, usable_cores
        n = sample or 16
        cores = usable_cores()
        torch.set_num_threads(cores)
        net = TRXOSOracle(self.state, self.L, self.J)
        sf = net.mlp(self.ss)
        q = self.q_host[:n]
        ref = {}
        med, total = median_time(lambda: ref.update(net.forward(None, self.way, q, ss_features=sf)), warm=1, iters=iters)
        # the parity half of the metric: the GPU's answers for the SAME windows against what was just timed
        logits, is_true, embed = self.eng.infer(q, want_embed=True)
        self.parity = ar_parity(logits, is_true, embed, ref)
        return {"value": round(n / med, 3), "unit": "windows/s", "cores": cores, "kind": "port",
                "sample": f"median of {iters} passes over {n} windows of {self.L}x{self.J} joints vs {self.way} classes, numpy "
                          f"oracle (support features cached), {total:.1f} s of CPU work"}

    def config(self, world):
        return {"workload": "BASELINE configs[2]: AR embed + tuple cross-attention match + open-set score, "
                            f"B={self.B} windows/GPU of {self.L}x{self.J} joints, way={self.way}",
                "per_gpu_batch": self.B, "seq_len": self.L, "n_joints": self.J, "way": self.way,
                "precision": self.precision,
                "parallelism": f"dp{world} (units sharded, one all-gather of results"
                               + (": RCCL behind the C ABI, isb_dist_all_gather)" if self.gather is not None else ")")}

    metric = "windows/sec (skeleton-window embed + few-shot match + open-set score)"
    unit = "windows/s"


def pick_workload(name):
    if name in ("auto", "pipeline", "hpe", "stream", "det"):
        import bench_workloads
        return bench_workloads.get(name)
    return ArWorkload


def main():
    args = parse()
    import torch

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))             # before ANY GPU call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs",
              file=sys.stderr)
        raise SystemExit(2)
    # stdout carries ONE JSON line and nothing else: librccl prints a version banner to file descriptor 1 when a communicator is created
    # (RCCL 2.26: "RCCL version : ..." and four more lines), which would stand in front of the line. From here on fd 1 IS stderr for
    # every library of this process; the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    # rehearsal of the N > 1 code path on a ONE-GPU box: ISB_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # ISB_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); never used by the driver
    if os.environ.get("ISB_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("ISB_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    # ISB_BENCH_FORCE_DIST=1: run the N > 1 code path (process group on RCCL, side-stream all-gather, barrier, all-reduce of the
    # time) with ONE rank -- the only way to exercise the real RCCL calls on a one-GPU box; never used by the driver
    force_dist = os.environ.get("ISB_BENCH_FORCE_DIST") == "1" and world == 1
    if force_dist:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
    dist_on = world > 1 or force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    t_gpu0 = time.perf_counter()
    args.dist_backend, args.force_dist = backend, force_dist
    W = pick_workload(args.workload)(args, rank, 2 if force_dist else world, local)      # (force: the workload takes its N > 1 branch)
    # set-up, not measurement: the first calls allocate the per-lane activation workspaces and load the code objects
    # (and, for N > 1, establish the RCCL rings), and a fraction of a second of load brings the clocks to their sustained
    # state (a 100-step run averages 19.9 ms per step, the first steps after start-up 21); the W warm-up steps the
    # caller asked for follow
    for _ in range(8 if args.workload in ("auto", "pipeline", "hpe", "ar") else 2):
        W.step()
    torch.cuda.synchronize()

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        W.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        W.step()
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # EVERY rank runs the profiling pass: its steps contain the step's all-gather, a collective that rank 0 alone
    # would wait on forever; only rank 0's numbers are reported
    roof = W.roofline(max(1, min(args.steps, 3)))
    barrier()
    # K steps in flight: `value` is overlapped throughput. The same steps one at a time through one engine and a step's latency under
    # overlap are first-class fields of the line (every rank steps: the steps hold the all-gather)
    in_flight = W.in_flight_report(args.steps) if hasattr(W, "in_flight_report") else None
    barrier()
    # N > 1: the gathered records of one step against an unsharded recomputation (rank 0), and RCCL's own count of the ranks
    gather_check, rccl_ranks = None, None
    if dist_on:
        g = getattr(W, "gather", None)
        if g is not None:
            rccl_ranks = g.rccl_ranks()
        if hasattr(W, "verify_gather"):
            gather_check = W.verify_gather(rank, 2 if force_dist else world) if not force_dist else None
        barrier()
    # extra measurements of the same run (N = 1): the fp32-grade AR precision, configs[3]'s whole 2048-frame batch
    extras = None
    if world == 1 and not force_dist and not args.no_extras and hasattr(W, "extras"):
        extras = W.extras(args)
    if world == 1 and args.min_gpu_seconds > 0:
        while time.perf_counter() - t_gpu0 < args.min_gpu_seconds:      # untimed: only keeps the GPU phase visible to samplers
            W.step()
            torch.cuda.synchronize()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = W.cpu_baseline(args.cpu_sample, max(1, args.cpu_iters))
    devices = [f"cuda:{local}"]
    if dist_on:
        import torch.distributed as dist
        world = dist.get_world_size()                    # what the process group says, not what the flags said
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank{rank}=cuda:{torch.cuda.current_device()}")
        devices = gathered

    if rank == 0 and roof is not None and hasattr(W, "step_flops"):
        # the WHOLE step against the dense 16-bit matrix peak: sum of the algorithmic FLOPs of every stage / the timed step
        sf = W.step_flops() / (dt / args.steps) / 1e12
        roof["step_achieved"] = round(sf, 2)
        roof["step_frac"] = round(sf / MFMA_PEAK_TFLOPS["bf16"], 4)
    if rank == 0:
        units = W.units_per_step() * world * args.steps
        line = {
            "metric": W.metric, "value": round(units / dt, 3), "unit": W.unit, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": W.precision if hasattr(W, "precision") else "bf16", "data": "synthetic",
            "config": dict(W.config(world), rccl_ranks=rccl_ranks, **(extras or {})), "roofline": roof, "cpu_baseline": cpu,
            # K > 1 steps in flight (config.steps_in_flight): `value` / `ms_per_step` are overlapped throughput, a step is FINISHED every
            # ms_per_step. in_flight.one_step_in_flight = the same steps one at a time through one engine (one isb_hpe_forward per step, the
            # library's two half-batch lanes), in_flight.step_latency_ms = a step's first launch -> its last kernel under overlap,
            # in_flight.engines = the engines share ONE device copy of the weights (isb_hpe_create_shared). null: one step at a time.
            "in_flight": in_flight,
            # error half of BASELINE.json's metric ("...; open-set score L2 vs ref"): GPU outputs against the CPU oracle
            # on the cpu_baseline sample (same inputs); null when the CPU leg is skipped (N > 1, --no-cpu-baseline)
            "parity": getattr(W, "parity", None),
            "world_size": world, "devices": devices,
            # N > 1: ranks in the library's RCCL communicator as ncclCommCount reports them (null: no RCCL collective in this run)
            # and the bit-for-bit check of the gathered records against an unsharded recomputation
            "rccl_ranks": rccl_ranks, "gather_check": gather_check,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
