"""CPU placement of the per-GPU ranks.

The reference pins every analyzer process to a core with ``taskset -p -c``
(radiotracking/__main__.py:122-128: core = device index mod cpu_count).  Here a
rank drives one GPU: it enqueues launches and polls pinned result memory the
kernels write over PCIe, so it should run on the cores of the NUMA node its GPU
hangs off.  :func:`plan` works out every rank's core set from sysfs alone
(``/sys/class/kfd`` for the GPUs in HIP's enumeration order, ``/sys/bus/pci``
for their NUMA node, ``/sys/devices/system/node`` for the node's cores) --
nothing here touches a GPU, so it can run in a fresh rank process before the
first HIP call (and must: the affinity mask is inherited by the runtime's
helper threads only if it is set before they start).  Ranks whose GPUs share a
node split its cores evenly into disjoint sets; without usable sysfs the cores
this process may run on are split evenly over the ranks.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """``"0-3,8,10-11"`` -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def gpu_pci_addresses(root: str = "/") -> List[str]:
    """PCI addresses (``dddd:bb:dd.f``) of the GPUs in KFD topology order -- the order HIP numbers them in when no
    ``*_VISIBLE_DEVICES`` variable re-orders them.  CPU nodes (no SIMDs) are skipped."""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        text = _read(os.path.join(base, str(n), "properties"))
        if not text:
            continue
        props: Dict[str, int] = {}
        for line in text.splitlines():
            k, _, v = line.partition(" ")
            try:
                props[k] = int(v)
            except ValueError:
                pass
        if props.get("simd_count", 0) <= 0:
            continue
        loc = props.get("location_id", 0)
        out.append(f"{props.get('domain', 0):04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}")
    return out


def visible_ordinals(n_gpus: int, env=os.environ) -> List[int]:
    """Topology indices behind HIP ordinals 0 .. under ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES lists of plain
    indices (UUID entries or anything else unparsable: identity).  ROCR_VISIBLE_DEVICES re-indexes what the runtime
    below HIP sees; HIP then applies ONE further list -- HIP_VISIBLE_DEVICES, or its alias CUDA_VISIBLE_DEVICES where
    that is unset: the two are never composed (a launcher that sets both to "1,0" means one swap, not two)."""
    order = list(range(n_gpus))
    hip_var = "HIP_VISIBLE_DEVICES" if env.get("HIP_VISIBLE_DEVICES") else "CUDA_VISIBLE_DEVICES"
    for var in ("ROCR_VISIBLE_DEVICES", hip_var):
        val = env.get(var)
        if not val:
            continue
        try:
            pick = [int(x) for x in val.split(",") if x.strip() != ""]
        except ValueError:
            return order
        if any(i < 0 or i >= len(order) for i in pick):
            return order
        order = [order[i] for i in pick]
    return order


def numa_node_of(pci: str, root: str = "/") -> int:
    text = _read(os.path.join(root, "sys/bus/pci/devices", pci, "numa_node"))
    try:
        return int(text.strip()) if text else -1
    except ValueError:
        return -1


def node_cpus(node: int, root: str = "/") -> List[int]:
    text = _read(os.path.join(root, f"sys/devices/system/node/node{node}/cpulist"))
    return parse_cpulist(text) if text else []


def core_groups(cpus: Sequence[int], root: str = "/") -> List[List[int]]:
    """`cpus` grouped by physical core (``topology/thread_siblings_list``): a core's hardware threads stay together, the
    groups in the order of their lowest thread.  Without that file every thread is a core of its own."""
    left = set(cpus)
    groups: List[List[int]] = []
    for c in sorted(left):
        if c not in left:
            continue
        text = _read(os.path.join(root, f"sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list"))
        try:
            sib = [x for x in parse_cpulist(text) if x in left] if text else []
        except ValueError:
            sib = []
        if c not in sib:
            sib = [c]
        groups.append(sorted(sib))
        left -= set(sib)
    return groups


def _split(cpus: Sequence[int], parts: int, which: int, root: str = "/") -> List[int]:
    """The `which`-th of `parts` disjoint shares of `cpus`, dealt by PHYSICAL core -- a core goes to one rank together with
    its SMT sibling (a contiguous split of "0-47,96-143" gave rank 0 the cores 0-23 and rank 2 their siblings 96-119:
    two ranks on the same physical cores).  Shares are contiguous runs of cores (sizes differ by at most one core; never
    empty while there are at least as many cores as parts -- with fewer, shares wrap around and overlap)."""
    groups = core_groups(cpus, root)
    if not groups:
        return []
    if len(groups) < parts:
        return list(groups[which % len(groups)])
    base, extra = divmod(len(groups), parts)
    lo = which * base + min(which, extra)
    return sorted(c for g in groups[lo:lo + base + (1 if which < extra else 0)] for c in g)


def plan(gpu_of_rank: Sequence[int], allowed: Sequence[int], root: str = "/", env=os.environ) -> List[dict]:
    """Core set of every rank: ``gpu_of_rank[r]`` = HIP ordinal rank r drives, ``allowed`` = the cores this job may use
    (``os.sched_getaffinity(0)``).  -> per rank ``{"cpus": [...], "numa_node": n | None, "pci": addr | None, "how": text}``.
    Deterministic: every rank computes the same plan and takes its own entry."""
    allowed = sorted(set(allowed))
    world = len(gpu_of_rank)
    gpus = gpu_pci_addresses(root)
    order = visible_ordinals(len(gpus), env)
    info = []
    for r, g in enumerate(gpu_of_rank):
        pci = gpus[order[g]] if gpus and 0 <= g < len(order) else None
        node = numa_node_of(pci, root) if pci else -1
        cpus = [c for c in node_cpus(node, root) if c in set(allowed)] if node >= 0 else []
        info.append((pci, node, cpus))
    out = []
    for r, (pci, node, cpus) in enumerate(info):
        if cpus:
            # ranks on the same node (several GPUs per socket, or several ranks on one GPU) share its cores evenly
            same = [q for q, (_, n2, c2) in enumerate(info) if n2 == node and c2]
            mine = _split(cpus, len(same), same.index(r), root)
            out.append({"cpus": mine, "numa_node": node, "pci": pci, "how": f"cores of NUMA node {node} (GPU {pci}), share {same.index(r) + 1} of {len(same)}"})
        else:
            # no NUMA information (a VM without it, numa_node = -1, sysfs hidden): an even split of what the job may use
            loose = [q for q, (_, _, c2) in enumerate(info) if not c2]
            taken = set(c for q, (_, _, c2) in enumerate(info) if c2 for c in c2)
            pool = [c for c in allowed if c not in taken] or allowed
            mine = _split(pool, len(loose), loose.index(r), root)
            out.append({"cpus": mine, "numa_node": None if node < 0 else node, "pci": pci,
                        "how": f"no NUMA information for GPU {pci}: share {loose.index(r) + 1} of {len(loose)} of the job's cores"})
    return out


def pin_rank(rank: int, gpu_of_rank: Sequence[int], root: str = "/") -> dict:
    """Pin THIS process to its share (``os.sched_setaffinity``) and return its plan entry (+ ``"pinned"``).  Call it
    before the first GPU call of the process.  Failing to pin is reported, not fatal (the reference only warns,
    __main__.py:127-128)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:  # not Linux
        return {"cpus": [], "numa_node": None, "pci": None, "how": "sched_getaffinity unavailable", "pinned": False}
    me = plan(gpu_of_rank, allowed, root)[rank]
    try:
        if me["cpus"]:
            os.sched_setaffinity(0, me["cpus"])
        me["pinned"] = bool(me["cpus"]) and sorted(os.sched_getaffinity(0)) == sorted(me["cpus"])
    except OSError as e:
        me["pinned"] = False
        me["how"] += f"; sched_setaffinity failed: {e}"
    return me
