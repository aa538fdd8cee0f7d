"""ctypes front end of helio_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhelio_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "helio_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libhelio_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def geometry(helios, sun, action, trig, target_pos, target_normal):
    """Returns actual [B,N,3], refl [B,N,3], inter [B,N,3], mask [B,N]."""
    helios, ph = _f(helios)
    sun, ps = _f(np.reshape(sun, (-1, 3)))
    B, N = sun.shape[0], helios.shape[0]
    action, pa = _f(np.reshape(action, (B, N, 3)))
    trig, pt = _f(np.reshape(trig, (B, N, 4)))
    tp, ptp = _f(target_pos)
    tn, ptn = _f(target_normal)
    outs = [np.empty((B, N, 3), np.float32) for _ in range(3)] + [np.empty((B, N), np.float32)]
    po = [o.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for o in outs]
    lib().oracle_geometry(ctypes.c_int(B), ctypes.c_int(N), ph, ps, pa, pt, ptp, ptn, *po)
    return tuple(outs)


def splat(inter, mask, helios, origin, u, v, xs, ys, sigma_scale):
    helios, ph = _f(helios)
    N = helios.shape[0]
    inter, pi = _f(np.reshape(inter, (-1, N, 3)))
    B = inter.shape[0]
    mask, pm = _f(np.reshape(mask, (B, N)))
    origin, po = _f(origin)
    u, pu = _f(u)
    v, pv = _f(v)
    xs, pxs = _f(xs)
    ys, pys = _f(ys)
    R = xs.shape[0]
    img = np.empty((B, R, R), np.float32)
    lib().oracle_splat(ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(R), pi, pm, ph, po, pu, pv,
                       pxs, pys, ctypes.c_float(sigma_scale),
                       img.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return img


def ideal_normals(helios, sun, target_pos):
    helios, ph = _f(helios)
    sun, ps = _f(np.reshape(sun, (-1, 3)))
    tp, pt = _f(target_pos)
    B, N = sun.shape[0], helios.shape[0]
    out = np.empty((B, N, 3), np.float32)
    lib().oracle_ideal_normals(ctypes.c_int(B), ctypes.c_int(N), ph, ps, pt,
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out
