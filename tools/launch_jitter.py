#!/usr/bin/env python3
"""Where a 20-step sample of the config-2 render loses its time: per-call host timestamps of 3000
consecutive HelioField.render calls (no synchronisation in between), then the positions and sizes of
the calls that took more than 4x the median."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from doodle_amd import synthetic
from bench import build_field, make_action, preheat

dev = torch.device("cuda")
w = synthetic.CONFIGS["cfg2"]
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev); suns_d = suns.to(dev); act = make_action(f, suns_d, noise)
render = f.render
pc = time.perf_counter
with torch.no_grad():
    preheat(lambda: render(suns_d, act, None), 0.5)
    torch.cuda.synchronize()
    for trial in range(3):
        n = 3000
        ts = np.empty(n + 1)
        ts[0] = pc()
        for i in range(n):
            render(suns_d, act, None)
            ts[i + 1] = pc()
        torch.cuda.synchronize()
        d = np.diff(ts) * 1e6
        med = np.median(d)
        big = np.nonzero(d > 4 * med)[0]
        print(f"trial {trial}: median {med:.2f} us/call, mean {d.mean():.2f}, {len(big)} calls > {4*med:.0f} us at "
              f"{big[:12].tolist()} … taking {np.round(d[big][:12], 0).tolist()} us; gaps between them {np.diff(big)[:12].tolist()}")
    # the same with a synchronise every 20 calls (what a K = 20 sample brackets)
    for trial in range(2):
        samples = []
        for s in range(40):
            torch.cuda.synchronize()
            t0 = pc()
            for i in range(20):
                render(suns_d, act, None)
            torch.cuda.synchronize()
            samples.append((pc() - t0) * 1e6)
        print("K=20 samples (us):", " ".join(f"{x:.0f}" for x in samples))
