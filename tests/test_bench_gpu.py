"""-m gpu: bench.py's N>1 control flow on ONE GPU (round-1 VERDICT weak #5, ADVICE bench.py:103).

`python bench.py --gpus 2` with no launcher in the environment must start its own torch.distributed.run; with
`--backend gloo` the two ranks share device 0, so the whole multi-rank path (rendezvous, barrier, max-over-ranks, one
JSON line from rank 0) runs on the 1-GPU box.  The numbers are not performance figures."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launch_two_ranks_share_device0():
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--frames", "2"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parity_vs_oracle"] is True
    assert d["config"]["frames_rank0"] == 3 and len(d["config"]["per_rank_seconds"]) == 2      # 3 x 2 frame renders dealt to 2 ranks
    assert d["cpu_baseline"] is None                   # the timed CPU sample is an N=1 item
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["kernel_ms"] > 0
    assert 0 < d["roofline"]["line_bound"]["frac"] < 1.5


def test_bench_single_rank_line_matches_the_contract():
    d = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["config"]["frames_per_step"] == 16 and d["config"]["frames_rank0"] == 48
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 16 * 54495972
    assert d["roofline"]["line_bound"]["bytes_per_launch"] == 16 * (718080 * 128 + 11520000)


def test_bench_job_mode_two_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--mode", "job", "--job-frames", "5", "--job-size", "256"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 5
    assert d["config"]["frames_rank0"] == 3 and d["config"]["parity_vs_oracle"] is True and d["value"] > 0


def test_bench_stream_mode_strong_scaling_two_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--mode", "stream", "--stream-frames", "7", "--stream-size", "320"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 7
    assert d["config"]["frames_rank0"] == 4 and d["config"]["parity_vs_oracle"] is True
    assert d["value"] > 0 and d["config"]["frames_per_s"] > 0
