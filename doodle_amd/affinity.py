"""Keep the launching host thread on the CPU socket the GPU hangs off.

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
