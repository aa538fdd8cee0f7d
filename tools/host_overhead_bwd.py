#!/usr/bin/env python3
"""Where the host time of render forward+backward goes at config 3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doodle_amd import synthetic
from bench import build_field, make_action

def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

w = synthetic.CONFIGS["cfg2"]
dev = torch.device("cuda")
helios, suns, errs, noise = synthetic.make_inputs(w, 0)
f = build_field(w, helios, errs, dev)
s = suns.to(dev); a = make_action(f, s, noise).requires_grad_(True)
G = torch.randn(w.B, w.R, w.R, device=dev); H = torch.randn(w.B, w.N, 3, device=dev)
print("forward with grad            %.1f us" % t(lambda: f.render(s, a, None)))
def fb_min():
    img, actual = f.render(s, a, None)
    torch.autograd.backward((img, actual), (G, H))
    a.grad = None
print("fwd + backward(img,actual)   %.1f us" % t(fb_min))
def fb_loss():
    img, actual = f.render(s, a, None)
    ((img * G).sum() + (actual * H).sum()).backward()
    a.grad = None
print("fwd + loss ops + backward    %.1f us" % t(fb_loss))
def fb_grad():
    img, actual = f.render(s, a, None)
    torch.autograd.grad((img * G).sum() + (actual * H).sum(), a)
print("fwd + loss ops + autograd.grad %.1f us" % t(fb_grad))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(200): fb_min()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=50))
