"""ShardedRenderer (doodle_amd/sharded.py) with the HIP kernels in separate PROCESSES — two and three ranks on the one
GPU of the box over gloo (RCCL refuses two ranks on one device; no multi-GPU box was ever available): every rank's
gathered images, geometry and own-row gradients must equal the unsharded render on the same GPU bit for bit, across
kernel regimes (the whole batch and a shard on its own would choose different kernels, forward and backward) — with
the gather over torch.distributed and over the opt-in peer-store transport (IPC-mapped receive buffers)."""
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
sr = ShardedRenderer(f, transport={transport!r})
assert sr.gather.transport == {transport!r}
for step in range({steps}):                   # (several steps: the peer-store transport alternates between two receive buffers)
    images, actual, refl = sr.render(suns, a, monitor=True, gather_geometry=True)
G = torch.randn(B, R, R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
(ga,) = torch.autograd.grad((images * G).sum() + actual.sum(), a)
torch.save({{"images": images.detach().cpu(), "actual": actual.detach().cpu(), "refl": refl.detach().cpu(), "grad": ga.cpu(),
            "rows": sr.local_rows(B)}}, os.path.join(out, f"rank{{rank}}.pt"))
sr.gather.close()
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,N,B,R,transport", [
    (2, 50, 25, 128, "torch"),       # config 2: the fused kernel for the batch and for a shard
    (2, 300, 40, 256, "torch"),      # batch: 128² register tiles; a 20-sun shard alone: another kernel
    (3, 1200, 7, 96, "torch"),       # ragged: 3 + 3 + 1 rows; few images of many heliostats
    # the opt-in peer-store transport (include/helio_comm.h helio_p2p_*): every rank stores its shard into the other ranks'
    # IPC-mapped receive buffers — here the other processes' buffers on the same GPU — three steps, three shard sizes
    (2, 50, 25, 128, "p2p"),
    (3, 301, 40, 100, "p2p"),        # ragged (14 + 14 + 12 rows); `actual` shards of 12642 floats: the dword form of the store kernel
])
def test_ranks_in_processes_equal_the_unsharded_render(tmp_path, world, N, B, R, transport):
    import socket
    from test_gpu_more import make_case
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, N=N, B=B, R=R, out=str(tmp_path), transport=transport,
                                    steps=3 if transport == "p2p" else 1))
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


def test_peer_store_gather_with_one_rank():
    """The p2p transport degenerates to a copy through its own receive buffer with one rank (no process group): the
    store kernel, the host-polled flag table, both parities, two shard sizes (one not a multiple of four floats)."""
    from doodle_amd.comm import ImageGather
    g = ImageGather(transport="p2p")
    assert g.transport == "p2p" and g.world == 1 and g.rccl_ranks is None
    try:
        for n in (4096, 1023):
            for step in range(3):
                x = torch.randn(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n + step))
                out = torch.full((n,), float("nan"), device="cuda")
                assert g.gather(x, out) is out and torch.equal(out, x)
        with pytest.raises(RuntimeError, match="float32 device"):
            g.gather(torch.zeros(4), torch.zeros(4))
    finally:
        g.close()
