"""CPU, world_size 2 over gloo: the N>1 path of bench.py / a sharded deployment -- units are split
by shard_range, each rank computes its slice (here with the oracle standing in for the HIP engine,
which needs a GPU), one all-gather of packed records, and the result must be BIT-identical to the
unsharded computation (no cross-rank arithmetic)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from isbfsar_amd import synth, weights
from isbfsar_amd.dist import all_gather_records, pack_records, shard_range, unpack_records


def test_shard_range_partitions():
    for n in (0, 1, 7, 256, 1023):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


L, J, WAY, B = 8, 6, 3, 7     # ragged on purpose: 7 windows over 2 ranks -> 4 + 3


def _compute(q):
    from oracle.ar_oracle import TRXOSOracle
    net = TRXOSOracle(weights.make_ar_state(L, J, seed=2), L, J)
    ss = synth.skeleton_windows(WAY, L, J, seed=50)
    out = net.forward(ss, WAY, q)
    return (torch.from_numpy(out["logits"]), torch.from_numpy(out["is_true"][:, 0]),
            torch.from_numpy(out["query_features"]))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q = synth.skeleton_windows(B, L, J, seed=60)
        counts = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
        a, b = shard_range(B, rank, world)
        rec = pack_records(*_compute(q[a:b]))
        full = all_gather_records(rec, counts)
        # equal-shard fast path too (first 6 windows -> 3 + 3)
        a2, b2 = shard_range(6, rank, world)
        full_eq = all_gather_records(pack_records(*_compute(q[a2:b2])))
        if rank == 0:
            ret["full"] = full.numpy()
            ret["full_eq"] = full_eq.numpy()
    finally:
        dist.destroy_process_group()


def test_sharded_equals_unsharded_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        full, full_eq = ret["full"], ret["full_eq"]
    torch.set_num_threads(1)
    q = synth.skeleton_windows(B, L, J, seed=60)
    ref = pack_records(*_compute(q)).numpy()
    # each rank computed its slice on its own; rows must match the unsharded run bit for bit
    ref_rows = np.concatenate([pack_records(*_compute(q[a:b])).numpy()
                               for a, b in (shard_range(B, r, world) for r in range(world))])
    assert np.array_equal(full, ref_rows)
    np.testing.assert_allclose(full, ref, rtol=0, atol=1e-6)      # BLAS may block differently per batch size
    logits, is_true, embed = unpack_records(torch.from_numpy(full), WAY, L)
    assert logits.shape == (B, WAY) and is_true.shape == (B,) and embed.shape == (B, L, 256)
    ref_eq = np.concatenate([pack_records(*_compute(q[a:b])).numpy()
                             for a, b in (shard_range(6, r, world) for r in range(world))])
    assert np.array_equal(full_eq, ref_eq)


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """bench.py must never report an N-GPU line from a run with another world size (round-1 behaviour: a silent N=1)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout


def test_bench_self_launch_starts_a_child_job():
    """Without a torchrun environment `--gpus 2` starts `python -m torch.distributed.run --nproc-per-node 2 bench.py ...`
    as a CHILD process and returns its exit code. No GPU here: the ranks stop at "needs a GPU" and the launcher must
    hand that failure through (non-zero), not print a line."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "launching 2 ranks" in r.stderr and "--nproc-per-node=2" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr and '"metric"' not in r.stdout
