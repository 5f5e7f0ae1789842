"""N>1 path on CPU: two processes over gloo shard frames with no data-path collective; the only
communication is bench.py's barrier + max-over-ranks of the elapsed time."""
import os
import socket
import sys

import numpy as np
import pytest

from gs360.sharding import frames_for_rank, shard_jobs


def test_frames_for_rank_partition():
    for n in (0, 1, 5, 16, 601):
        for world in (1, 2, 4, 8):
            parts = [frames_for_rank(n, world, r) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        frames_for_rank(4, 2, 2)


def test_shard_jobs_keeps_views_of_a_frame_together():
    jobs = [(["x"], f"f{i}.png", f"f{i}_{v}.jpg") for i in range(5) for v in "ABC"]
    parts = [shard_jobs(jobs, 2, r) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == sorted(jobs)
    for p in parts:
        srcs = {j[1] for j in p}
        assert all(sum(1 for j in p if j[1] == s) == 3 for s in srcs)
    assert {j[1] for j in parts[0]}.isdisjoint({j[1] for j in parts[1]})


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import orc
    # every rank renders ITS frames (oracle stands in for the GPU here); results are a pure function of
    # (frame, view), so the gathered set must equal the single-process result
    n_frames = 5
    rng = np.random.default_rng(123)
    frames = [rng.integers(0, 256, (32, 64, 3), dtype=np.uint8) for _ in range(n_frames)]
    views = [orc.make_view(y, 0, 90, 90, 16, 16) for y in (0, 120, -120)]
    mine = frames_for_rank(n_frames, world, rank)
    sums = torch.zeros(n_frames, dtype=torch.int64)
    for f in mine:
        outs = orc.equirect_views_u8(frames[f], views)
        sums[f] = int(sum(int(o.astype(np.int64).sum()) for o in outs))
    dist.barrier()
    elapsed = torch.tensor([0.25 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)          # bench.py: max over ranks
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)             # test-only gather of the checksums
    q.put((rank, mine, sums.tolist(), float(elapsed.item())))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, sums0, e0), (r1, mine1, sums1, e1) = res
    assert sorted(mine0 + mine1) == list(range(5)) and set(mine0).isdisjoint(mine1)
    assert sums0 == sums1 and all(v > 0 for v in sums0)
    assert e0 == e1 == 1.25
    # single-process truth
    from oracle import orc
    rng = np.random.default_rng(123)
    frames = [rng.integers(0, 256, (32, 64, 3), dtype=np.uint8) for _ in range(5)]
    views = [orc.make_view(y, 0, 90, 90, 16, 16) for y in (0, 120, -120)]
    truth = [int(sum(int(o.astype(np.int64).sum()) for o in orc.equirect_views_u8(f, views))) for f in frames]
    assert truth == sums0


def test_bench_world_size_mismatch_is_refused():
    """ADVICE r1 (bench.py:103): --gpus N with a launcher that started another world size exits non-zero before any GPU call"""
    import pathlib
    import subprocess
    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr
