"""bench.py end to end on the GPU box: the one-rank line the driver parses, and the N > 1 control flow — which no
multi-GPU box has ever run — rehearsed with two ranks on the one GPU (`--rehearse`: gloo, every rank on cuda:0)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    return json.loads(lines[0])


def test_one_rank_line_meets_the_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu", "--no-large",
                        "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert all(k in d for k in CONTRACT) and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] > 1e6 and d["unit"] == "frames/s" and d["config"]["workload"].startswith("N=50,B=25,R=128")
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] <= 1 and "rehearsal" not in d


def test_two_ranks_walk_the_multi_gpu_control_flow():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--rehearse",
           "--no-cpu", "--cfg5-suns", "8", "--cfg5-steps", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)                       # rank 0 alone prints, and prints one line
    assert all(k in d for k in CONTRACT) and d["n_gpus"] == 2 and d["rehearsal"] is True
    assert d["config"]["global_batch"] == 50 and d["scaling"] == "weak" and d["value"] > 0
    assert d["collective"]["world_size"] == 2
    assert d["with_all_gather_every_step"]["frames_per_s"] > 0          # the gathered loop ran on both ranks
    shard = d["multi_gpu_of_record"]
    assert "error" not in shard and shard["n_gpus"] == 2 and shard["frames_per_s"] > 0
    assert d["scaling_of_record"]["frames_per_s"] == shard["frames_per_s"]
