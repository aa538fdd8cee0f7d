"""Sun-batch sharding (doodle_amd/sharded.py): CPU, gloo, world_size 2 — plus the
single-process statement of the same property.  The sharded result must equal the
unsharded one BIT FOR BIT (images are independent per sun)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden
import oracle_backend
from test_host_logic import field_from


def test_shard_rows_cover_the_batch():
    from doodle_amd.sharded import shard_rows
    for B in (1, 2, 7, 25, 26, 4096):
        for world in (1, 2, 3, 4, 8):
            rows = [shard_rows(B, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == B
            assert all(rows[r][1] == rows[r + 1][0] for r in range(world - 1))
            assert all(b1 - b0 <= chunk for b0, b1, chunk in rows)
            assert len({chunk for _, _, chunk in rows}) == 1


def test_render_rows_equals_full_render(monkeypatch):
    oracle_backend.install(monkeypatch)
    g = golden("g8_ragged_n201_b7_r48")
    f = field_from(g)
    sun, act = torch.from_numpy(g["sun"]), torch.from_numpy(g["action"])
    full, actual, refl = f.render(sun, act, None, monitor=True)
    for b0, b1 in ((0, 3), (3, 4), (4, 7)):
        img, a, r = f.render_rows(sun[b0:b1], act[b0:b1], b0, 7, monitor=True)
        assert torch.equal(img, full[b0:b1]) and torch.equal(a, actual[b0:b1])
        assert torch.equal(r, refl.view(7, -1, 3)[b0:b1].reshape(-1, 3))
    assert np.array_equal(full.numpy(), g["image"]) or np.allclose(full.numpy(), g["image"], rtol=1e-5, atol=1e-8)


def _worker(rank, world, port, out_dir, nsuns):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import doodle_amd.field as field_mod
        ops = oracle_backend.OracleOps()
        field_mod._get_ops = lambda: ops
        from doodle_amd.sharded import ShardedRenderer
        g = golden("g8_ragged_n201_b7_r48")          # B = 7: ragged over 2, 3 and 4 ranks
        f = field_from(g)
        sun = torch.from_numpy(g["sun"])[:nsuns]
        act = torch.from_numpy(g["action"])[:nsuns].clone().requires_grad_(True)
        sr = ShardedRenderer(f)
        images, actual, refl = sr.render(sun, act, monitor=True, gather_geometry=True)
        G = torch.from_numpy(g["G"])[:nsuns]
        (ga,) = torch.autograd.grad((images * G).sum(), act)   # replicated loss; no backward collective
        b0, b1 = sr.local_rows(nsuns)
        # second call with the same inputs: sharded renders stay deterministic
        images2, _ = sr.render(sun, act.detach())
        torch.save({"images": images.detach(), "actual": actual.detach(), "refl": refl.detach(),
                    "grad": ga, "rows": (b0, b1), "same": torch.equal(images.detach(), images2)},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nsuns,rows", [
    (2, 7, [(0, 4), (4, 7)]),
    (3, 7, [(0, 3), (3, 6), (6, 7)]),
    (4, 7, [(0, 2), (2, 4), (4, 6), (6, 7)]),
    (3, 2, [(0, 1), (1, 2), (2, 2)]),          # more ranks than suns: the last rank renders nothing (padding only)
    (4, 5, [(0, 2), (2, 4), (4, 5), (5, 5)]),
])
def test_gloo_ranks_bit_identical_to_unsharded(tmp_path, monkeypatch, world, nsuns, rows):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), nsuns), nprocs=world, join=True)
    oracle_backend.install(monkeypatch)
    g = golden("g8_ragged_n201_b7_r48")
    f = field_from(g)
    act = torch.from_numpy(g["action"])[:nsuns].clone().requires_grad_(True)
    full, actual, refl = f.render(torch.from_numpy(g["sun"])[:nsuns], act, None, monitor=True)
    (gfull,) = torch.autograd.grad((full * torch.from_numpy(g["G"])[:nsuns]).sum(), act)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert [o["rows"] for o in outs] == rows
    total = torch.zeros_like(gfull)
    for o in outs:
        assert o["same"]
        assert torch.equal(o["images"], full.detach())             # every rank holds the full batch
        assert torch.equal(o["actual"], actual.detach()) and torch.equal(o["refl"], refl.detach())
        b0, b1 = o["rows"]
        assert torch.equal(o["grad"][b0:b1], gfull[b0:b1])         # own rows: full gradient (per rank)
        other = torch.cat([o["grad"][:b0], o["grad"][b1:]])
        assert other.numel() == 0 or float(other.abs().max()) == 0.0   # foreign rows: none (no collective)
        total += o["grad"]
    assert torch.equal(total, gfull)                               # the ranks' gradients tile the batch exactly


def test_image_gather_never_changes_transport_silently(monkeypatch):
    """A failed RCCL set-up raises — it does not fall back to torch.distributed with a warning: on a multi-GPU run
    that would change what bench.py measures, with ``rccl_ranks: null`` as the only trace."""
    from doodle_amd import comm

    def broken(*a, **k):
        raise OSError("libhelio_comm.so: cannot open shared object file")
    monkeypatch.setattr(comm, "load_comm_library", broken)
    with pytest.raises(RuntimeError, match="HELIO_COMM=torch"):
        comm.ImageGather(transport="rccl")
    monkeypatch.setenv("HELIO_COMM", "rccl")                       # what "auto" resolves to on GPUs with an nccl group
    with pytest.raises(RuntimeError, match="could not be set up"):
        comm.ImageGather(transport="auto")
    monkeypatch.delenv("HELIO_COMM")
    with pytest.raises(ValueError, match="unknown transport"):
        comm.ImageGather(transport="mpi")
    g = comm.ImageGather(transport="auto")                         # no GPU, no process group: torch transport, by rule
    assert g.transport == "torch" and g.rccl_ranks is None
    out = torch.empty(6)
    g.gather(torch.arange(6.0), out)
    assert torch.equal(out, torch.arange(6.0))
