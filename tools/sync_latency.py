#!/usr/bin/env python3
"""What the end-of-region fence of a K = 20 bench sample costs: wall time of K config-2 renders +
fence, for (a) torch.cuda.synchronize(), (b) an event polled with query() and then synchronize()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import synthetic
from bench import build_field, make_action, preheat

dev = torch.device("cuda")
w = synthetic.CONFIGS["cfg2"]
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
render = f.render
ev = torch.cuda.Event()


def sample(K, fence):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        render(suns_d, act, None)
    fence()
    return (time.perf_counter() - t0) * 1e6


def spin():
    ev.record()
    while not ev.query():
        pass
    torch.cuda.synchronize()


with torch.no_grad():
    preheat(lambda: render(suns_d, act, None), 0.5)
    torch.cuda.synchronize()
    first = [sample(20, torch.cuda.synchronize) for _ in range(6)]
    print("K=20 samples in order, right after a 0.5 s preheat + synchronize:", " ".join(f"{x:.1f}" for x in first))
    import gc
    gc.disable()
    preheat(lambda: render(suns_d, act, None), 0.5)
    torch.cuda.synchronize()
    first = [sample(20, torch.cuda.synchronize) for _ in range(6)]
    gc.enable()
    print("the same with the cycle collector off:", " ".join(f"{x:.1f}" for x in first))
    for K in (1, 20, 200, 2000):
        for name, fence in (("synchronize", torch.cuda.synchronize), ("event.query spin + synchronize", spin)):
            xs = sorted(sample(K, fence) for _ in range(15))
            print(f"K={K:5d} {name:32s} median {xs[7]:9.1f} us  min {xs[0]:9.1f} us  → {xs[7]/K:6.2f} us/step")
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"synchronize() on an idle device: {(t1-t0)*1e6:.1f} us")
