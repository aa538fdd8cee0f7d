"""bench.py's time-based preheat with a process group, deterministically (CPU, gloo, world_size 2 and 3).

Every burst of the preheat ends in a collective (the fence), so all ranks must run the SAME number of bursts.  Round 3's
two-rank rehearsal hung one run in six: each rank ended the loop by its own clock, and whenever a burst ended within the
ranks' start skew of the deadline one rank ran one barrier more.  Here the ranks get INJECTED clocks that disagree about
the deadline by construction — rank r's clock advances (1 + 0.3 r) seconds per reading — so that the rank-local rule
would give 4, 3 and 3 bursts; the agreed rule (an all-reduce MIN of each rank's verdict per burst, bench.preheat) must
give every rank the same count, and every fence must pair up (a mis-paired barrier would hang: the spawn has a deadline).
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class SkewedClock:
    """A wall clock that advances ``step`` seconds per reading."""

    def __init__(self, step):
        self.t, self.step = 0.0, step

    def __call__(self):
        self.t += self.step
        return self.t - self.step


def local_rule_bursts(step, seconds):
    """What a rank would run if it ended the loop by its own clock (round 3's bug)."""
    clock, n = SkewedClock(step), 0
    t0 = clock()
    while clock() - t0 < seconds:
        n += 1
    return n


def _worker(rank, world, port, out_dir, seconds):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import bench
        calls, fences = [0], [0]

        def fn():
            calls[0] += 1

        def fence():
            dist.barrier()
            fences[0] += 1

        n = bench.preheat(fn, seconds, burst=5, fence=fence, dist=dist, device="cpu", clock=SkewedClock(1.0 + 0.3 * rank))
        # one more collective after the loop: it pairs with the other ranks' only if nobody ran an extra barrier
        total = torch.tensor([n], dtype=torch.int32)
        dist.all_reduce(total)
        torch.save({"bursts": n, "calls": calls[0], "fences": fences[0], "total": int(total.item())},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_ranks_run_the_same_number_of_preheat_bursts(tmp_path, world):
    seconds = 3.5
    alone = [local_rule_bursts(1.0 + 0.3 * r, seconds) for r in range(world)]
    assert len(set(alone)) > 1, alone                  # the injected clocks DO disagree: the local rule would mis-pair
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), seconds), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    counts = {o["bursts"] for o in outs}
    assert counts == {min(alone)}, (outs, alone)       # the slowest clock's verdict ends the loop for everybody
    for o in outs:
        assert o["calls"] == 5 * o["bursts"] and o["fences"] == o["bursts"] and o["total"] == world * o["bursts"]


def test_preheat_without_a_process_group_runs_on_its_own_clock():
    sys.path.insert(0, ROOT)
    import bench
    calls = [0]
    n = bench.preheat(lambda: calls.__setitem__(0, calls[0] + 1), 3.5, clock=SkewedClock(1.0))
    assert n == local_rule_bursts(1.0, 3.5) == calls[0]
