"""ShardedRenderer (doodle_amd/sharded.py) with the HIP kernels in separate PROCESSES — two and three ranks on the one
GPU of the box over gloo (RCCL refuses two ranks on one device; no multi-GPU box was ever available): every rank's
gathered images, geometry and own-row gradients must equal the unsharded render on the same GPU bit for bit, across
kernel regimes (the whole batch and a shard on its own would choose different forward kernels)."""
import os
import signal
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from test_gpu_more import make_case
N, B, R, out = {N}, {B}, {R}, {out!r}
dist.init_process_group("gloo")
rank = dist.get_rank()
from doodle_amd.sharded import ShardedRenderer
f, _, suns, _, act = make_case(N=N, B=B, R=R, sigma=0.01, err=90.0, seed=3, span=30.0)
a = act.to("cuda").requires_grad_(True)
sr = ShardedRenderer(f, transport="torch")
images, actual, refl = sr.render(suns, a, monitor=True, gather_geometry=True)
G = torch.randn(B, R, R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
(ga,) = torch.autograd.grad((images * G).sum() + actual.sum(), a)
torch.save({{"images": images.detach().cpu(), "actual": actual.detach().cpu(), "refl": refl.detach().cpu(), "grad": ga.cpu(),
            "rows": sr.local_rows(B)}}, os.path.join(out, f"rank{{rank}}.pt"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,N,B,R", [(2, 50, 25, 128),       # config 2: the fused kernel for the batch and for a shard
                                         (2, 300, 40, 256),      # batch: 128² register tiles; a 20-sun shard alone: another kernel
                                         (3, 1200, 7, 96)])      # ragged: 3 + 3 + 1 rows; few images of many heliostats
def test_ranks_in_processes_equal_the_unsharded_render(tmp_path, world, N, B, R):
    import socket
    from test_gpu_more import make_case
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, N=N, B=B, R=R, out=str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, start_new_session=True)
    try:
        _, err = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        _, err = p.communicate()
        pytest.fail("ranks hung; stderr tail:\n" + err[-3000:])
    assert p.returncode == 0, err[-3000:]

    f, _, suns, _, act = make_case(N=N, B=B, R=R, sigma=0.01, err=90.0, seed=3, span=30.0)
    a = act.to("cuda").requires_grad_(True)
    full, actual, refl = f.render(suns, a, None, monitor=True)
    G = torch.randn(B, R, R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    (gfull,) = torch.autograd.grad((full * G).sum() + actual.sum(), a)
    total = torch.zeros_like(gfull).cpu()
    covered = 0
    for r in range(world):
        o = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert torch.equal(o["images"], full.detach().cpu())           # every rank holds the whole batch
        assert torch.equal(o["actual"], actual.detach().cpu()) and torch.equal(o["refl"], refl.detach().cpu())
        b0, b1 = o["rows"]
        covered += b1 - b0
        assert torch.equal(o["grad"][b0:b1], gfull[b0:b1].cpu())       # own rows: the full gradient, no collective
        other = torch.cat([o["grad"][:b0], o["grad"][b1:]])
        assert other.numel() == 0 or float(other.abs().max()) == 0.0
        total += o["grad"]
    assert covered == B and torch.equal(total, gfull.cpu())
