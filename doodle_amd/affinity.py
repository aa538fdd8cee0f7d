"""Keep the launching host thread near the GPU — and near the runtime's own threads.

A config-2 render is 3.7 µs of GPU work behind a ≈3.4 µs HIP launch: the loop is bound by the host thread's
doorbell and signal traffic to the card.  On a two-socket MI355X host (2 × 64 cores, the eight GPUs split over
the sockets) a process the scheduler happens to place on the far socket pays every one of those PCIe / xGMI
writes across the inter-socket link — 10–20 % more time per launch-bound step
(tools/sync_latency.py under ``taskset``).  ``bind_to_gpu_node`` restricts the calling process to the CPUs of
the NUMA node the device reports in sysfs; one process per GPU, so each rank binds to its own card's node.
Nothing here touches the device or the numbers: it is ``numactl --cpunodebind`` from inside the process.
"""
from __future__ import annotations

import os
from typing import Optional


def _parse_cpulist(text: str) -> set:
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index: int = 0) -> Optional[int]:
    """NUMA node of HIP device ``device_index`` (None when sysfs does not say, or says -1)."""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    try:
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
            node = int(fh.read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def node_cpus(node: int) -> set:
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            return _parse_cpulist(fh.read())
    except OSError:
        return set()


def bind_to_gpu_node(device_index: int = 0) -> Optional[dict]:
    """Restrict this process to the CPUs of the device's NUMA node (within its current affinity mask).
    → {"numa_node": n, "cpus": count} or None when nothing was changed (no sysfs entry, single node,
    HELIO_NUMA_BIND=0, or the intersection with the current mask is empty)."""
    if os.environ.get("HELIO_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    node = gpu_numa_node(device_index)
    if node is None:
        return None
    allowed = os.sched_getaffinity(0)
    cpus = node_cpus(node) & allowed
    if not cpus or cpus == allowed:
        return None
    os.sched_setaffinity(0, cpus)
    return {"numa_node": node, "cpus": len(cpus)}


def _l3_groups(cpus: set) -> list:
    """The sets of CPUs that share a last-level cache (a CCD on EPYC), restricted to ``cpus``, in CPU order."""
    seen, groups = set(), []
    for c in sorted(cpus):
        if c in seen:
            continue
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list") as fh:
                grp = _parse_cpulist(fh.read()) & cpus
        except OSError:
            grp = {c}
        if not grp:
            grp = {c}
        seen |= grp
        groups.append(grp)
    return groups


def _set_all_threads(cpus: set) -> None:
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
        except (OSError, ValueError):
            pass                                   # a thread that ended meanwhile


def gpu_slot_on_node(device_index: int) -> tuple:
    """→ (index of the device among the visible devices of its NUMA node, how many there are)."""
    import torch
    node = gpu_numa_node(device_index)
    same = [d for d in range(torch.cuda.device_count()) if gpu_numa_node(d) == node]
    return (same.index(device_index) if device_index in same else 0), max(len(same), 1)


def widen_to_node(device_index: int = 0) -> None:
    """Every thread of the process back on ALL CPUs of the device's NUMA node (within the process's mask when it
    was bound): for phases that are not launch-bound and run helper threads of their own (RCCL's proxies)."""
    if os.environ.get("HELIO_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return
    node = gpu_numa_node(device_index)
    cpus = node_cpus(node) if node is not None else set()
    if cpus:
        _set_all_threads(cpus)


def bind_to_gpu_ccd(device_index: int = 0, slot: Optional[int] = None) -> Optional[dict]:
    """Restrict EVERY thread of this process to ONE last-level-cache group (CCD) of the device's NUMA node.

    Measured (tools/core_sweep.py, config-2 loop, K = 20 renders + fence, main thread pinned core by core):
    the cores of exactly one CCD give 90 µs, all others of the same NUMA node 120 µs — and WHICH CCD differs
    from process to process: it is the one the HIP / ROCr runtime's helper threads happened to start on (every
    launch hands cache lines — queue pointers, completion signals — between them and the launching thread;
    inside one L3 that is cheap).  So the binding has to cover the runtime's threads too, not just the caller:
    call this AFTER the device is initialised (the threads exist) and once; threads created later inherit it.
    ``slot`` spreads several processes of one node over its CCDs; by default the device's index among the
    devices of ITS NUMA node (local ranks 0 and 4 of an 8-GPU, 2-node host both take the first CCD of their own
    node).  With more GPUs than CCDs on a node two ranks share one: reported as ``shared_ccd``, with a warning.
    → {"numa_node", "cpus", "l3_group", "shared_ccd"} or None when nothing was changed."""
    if os.environ.get("HELIO_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    node = gpu_numa_node(device_index)
    allowed = os.sched_getaffinity(0)
    cpus = (node_cpus(node) & allowed) if node is not None else set(allowed)
    if not cpus:
        return None
    groups = _l3_groups(cpus)
    if not groups:
        return None
    gpus_here = 1
    if slot is None:
        slot, gpus_here = gpu_slot_on_node(device_index)
    shared = gpus_here > len(groups)
    if shared:
        import warnings
        warnings.warn(f"doodle_amd.affinity: {gpus_here} GPUs on NUMA node {node} but {len(groups)} last-level-cache groups: "
                      "some ranks share one", stacklevel=2)
    grp = groups[slot % len(groups)]
    _set_all_threads(grp)
    return {"numa_node": node, "cpus": len(grp), "l3_group": f"{min(grp)}-{max(grp)}", "shared_ccd": shared}


def restore(mask: set) -> None:
    """Put every thread of the process back on ``mask`` (what ``os.sched_getaffinity(0)`` said before binding)."""
    if hasattr(os, "sched_setaffinity") and mask:
        _set_all_threads(set(mask))
