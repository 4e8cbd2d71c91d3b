"""Keeps the host threads that drive a GPU on cores that share one L3 cache.

A frame of the rasteriser is ~40 kernel launches issued by TWO threads -- the caller's (forward) and PyTorch's autograd
thread of the device (backward) -- next to the HIP runtime's helper threads; they hand work to each other through
condition variables.  An un-pinned process on a large host has them scattered over core complexes and sockets, and
every hand-over then costs a cross-complex (or cross-socket) wake-up.  Frames that are bound by the host feel it
(MI355X box, 2 x EPYC 9575F, 256 hardware threads, bench.py means of 50 steps): 10k Gaussians at 256 x 256
0.48-0.54 ms un-pinned, 0.53 with one core on each socket, **0.29 on the cores of one complex**; 1e5 Gaussians at 800 x
800 0.43-0.54 -> 0.33 ms.  Frames bound by the GPU (the 1e6-Gaussian headline) do not care.  PyTorch's intra-op thread
pool is cut to the complex's cores at the same time (it was sized for the whole machine).

Nothing here runs unless it is called: ``bench.py``, the two command lines and the benchmark scripts call
``pin_host_threads`` once at start-up (``GS_PIN_HOST_THREADS=0`` turns it off); an application embedding the operator
decides for itself.  Linux only (sysfs + sched_setaffinity); anywhere else, or when the topology cannot be read, it
does nothing and says so by returning None.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Set

_original_mask: Optional[Set[int]] = None   # the process's mask before the first pin (data-loader workers get it back)
_original_torch_threads: Optional[int] = None


def _parse_cpu_list(text: str) -> Set[int]:
    cpus: Set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def gpu_local_cpus(device_index: int) -> Optional[Set[int]]:
    """CPUs of the NUMA node the GPU hangs off (sysfs ``local_cpulist`` of its PCI function), or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None
    text = _read(f"/sys/bus/pci/devices/{bdf}/local_cpulist")
    cpus = _parse_cpu_list(text) if text else set()
    return cpus or None


def l3_groups(cpus: Set[int]) -> List[Set[int]]:
    """The given CPUs grouped by the L3 cache they share (sysfs cache/index3/shared_cpu_list), lowest CPU first."""
    groups: Dict[frozenset, Set[int]] = {}
    for c in sorted(cpus):
        text = _read(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list")
        if not text:
            return []
        shared = frozenset(_parse_cpu_list(text))
        groups.setdefault(shared, set()).add(c)
    return sorted(groups.values(), key=min)


def physical_cores(cpus: Set[int]) -> int:
    """Number of cores behind the given hardware threads (SMT siblings counted once)."""
    cores = set()
    for c in cpus:
        text = _read(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")
        cores.add(frozenset(_parse_cpu_list(text)) if text else frozenset([c]))
    return max(len(cores), 1)


def _all_thread_ids() -> List[int]:
    try:
        return [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        return [0]


def pin_host_threads(device_index: int = 0, slot: Optional[int] = None) -> Optional[Set[int]]:
    """Restrict every thread of this process (those that exist now; later ones inherit) to the cores of ONE L3 complex on
    the GPU's NUMA node.  ``slot`` picks the complex (default: LOCAL_RANK when a launcher set it, else the device index:
    the ranks of a node, or independent jobs on different GPUs, spread over the complexes).  Returns the CPU set, or None
    when nothing was changed."""
    global _original_mask
    if os.environ.get("GS_PIN_HOST_THREADS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = set(os.sched_getaffinity(0))
    if _original_mask is None:
        _original_mask = set(allowed)
    near = gpu_local_cpus(device_index)
    candidates = (near & allowed) if near else allowed
    groups = [g for g in l3_groups(candidates or allowed) if g]
    if not groups:
        return None
    if slot is None:
        rank = os.environ.get("LOCAL_RANK", "")
        slot = int(rank) if rank.isdigit() else _stable_device_slot(device_index)
    chosen = groups[slot % len(groups)]
    if os.environ.get("GS_PIN_HOST_THREADS_VERBOSE") == "1":
        print(f"[host_affinity] device {device_index}: slot {slot}, cpus {sorted(chosen)}", flush=True)
    if len(chosen) >= len(allowed):
        return None   # already that narrow
    for tid in _all_thread_ids():
        try:
            os.sched_setaffinity(tid, chosen)
        except OSError:
            pass      # a thread that ended meanwhile
    # PyTorch sized its intra-op (OpenMP) pool for the whole machine; a parallel region of 256 threads squeezed onto one
    # complex is worse than no pinning at all (measured: a 2,000-iteration training run 4.9 s un-pinned, 10.0 s pinned
    # with the pool left alone, 2.4 s pinned with the pool cut to the complex)
    global _original_torch_threads
    try:
        import torch
        if _original_torch_threads is None:
            _original_torch_threads = torch.get_num_threads()
        torch.set_num_threads(max(1, min(_original_torch_threads, physical_cores(chosen))))
    except Exception:
        pass
    return chosen


def _stable_device_slot(device_index: int) -> int:
    """A slot for a process that was given no LOCAL_RANK: the PHYSICAL device's identity, not its index in this process.
    Devices are narrowed with HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, or by a container that shows
    every pod its one GPU as "0" (round 6: four pods of one host, each with HIP_VISIBLE_DEVICES=0 in its own device namespace,
    all took the first complex of the same socket -- CPUs 0-7 -- and stalled one another's host threads for milliseconds at
    a time): so the device's PCI bus number first -- what the hardware says --, then the position the visibility list names,
    then the index."""
    try:
        import torch
        bus = getattr(torch.cuda.get_device_properties(device_index), "pci_bus_id", None)
        if bus is not None:
            return int(bus)
    except Exception:
        pass
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        text = os.environ.get(var, "")
        if text:
            entries = [e.strip() for e in text.split(",") if e.strip()]
            if 0 <= device_index < len(entries) and entries[device_index].isdigit():
                return int(entries[device_index])
            break
    return int(device_index)


def unpin_host_threads() -> None:
    """Give every thread of the process the mask it had before the first ``pin_host_threads`` (for CPU-parallel work:
    the oracle leg of bench.py runs on all cores)."""
    if _original_mask is None or not hasattr(os, "sched_setaffinity"):
        return
    for tid in _all_thread_ids():
        try:
            os.sched_setaffinity(tid, _original_mask)
        except OSError:
            pass
    if _original_torch_threads is not None:
        try:
            import torch
            torch.set_num_threads(_original_torch_threads)
        except Exception:
            pass


def original_mask() -> Optional[Set[int]]:
    """The affinity mask the process had before it was pinned (None: never pinned)."""
    return None if _original_mask is None else set(_original_mask)


def reset_worker_affinity(_worker_id: int = 0, mask: Optional[Set[int]] = None) -> None:
    """``worker_init_fn`` for ``torch.utils.data.DataLoader``: worker processes are started from the pinned main thread and
    would all sit on its complex.  Forked workers inherit this module's record of the original mask; spawned / fork-server
    workers do not -- pass it explicitly (``functools.partial(reset_worker_affinity, mask=original_mask())``); without either
    the allowed set of the cgroup (``Cpus_allowed_list`` of /proc/self/status) is taken."""
    if not hasattr(os, "sched_setaffinity"):
        return
    target = mask if mask is not None else _original_mask
    if target is None:
        text = _read("/proc/self/status") or ""
        for line in text.splitlines():
            if line.startswith("Cpus_allowed_list:"):
                target = set(_parse_cpu_list(line.split(":", 1)[1].strip()))
    if target:
        try:
            os.sched_setaffinity(0, target)
        except OSError:
            pass
