#!/usr/bin/env python3
"""Does the launch-bound loop depend on WHICH core of the GPU's NUMA node launches?  Pins the main thread to one
core at a time (every 4th core of the node), times K = 20 config-2 renders + fence, prints median / min per core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import affinity, synthetic
from bench import build_field, make_action, preheat

dev = torch.device("cuda")
node = affinity.gpu_numa_node(0)
cpus = sorted(affinity.node_cpus(node) & os.sched_getaffinity(0)) if node is not None else sorted(os.sched_getaffinity(0))
w = synthetic.CONFIGS["cfg2"]
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
render = f.render

def sample(K=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        render(suns_d, act, None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6

print(f"GPU NUMA node {node}: {len(cpus)} cpus ({cpus[0]}..{cpus[-1]})")
with torch.no_grad():
    preheat(lambda: render(suns_d, act, None), 0.3)
    for rnd in range(2):
        row = []
        for c in cpus[::4][:32]:
            os.sched_setaffinity(0, {c})
            for _ in range(3):
                sample()
            xs = sorted(sample() for _ in range(9))
            row.append(f"{c}:{xs[4]:.0f}/{xs[0]:.0f}")
        print(f"round {rnd}: core:median/min us  " + " ".join(row), flush=True)
    os.sched_setaffinity(0, set(cpus))
    xs = sorted(sample() for _ in range(15))
    print(f"whole node mask: median {xs[7]:.0f} min {xs[0]:.0f}")
