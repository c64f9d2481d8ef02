"""CPU placement of the per-GPU ranks (pyradiotracking_amd/affinity.py): the plan is worked out from sysfs alone, so it
is tested here on fabricated trees (the reference pins each analyzer with ``taskset``, radiotracking/__main__.py:122-128)."""
import os

from pyradiotracking_amd import affinity


def _tree(root, gpus, nodes):
    """gpus: [(domain, bus, dev, fn, numa)] in KFD order behind one CPU node; nodes: {node: cpulist text}"""
    base = root / "sys/class/kfd/kfd/topology/nodes"
    (base / "0").mkdir(parents=True)
    (base / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (dom, bus, dev, fn, numa) in enumerate(gpus, start=1):
        (base / str(i)).mkdir()
        (base / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {(bus << 8) | (dev << 3) | fn}\ndomain {dom}\n")
        pci = root / "sys/bus/pci/devices" / f"{dom:04x}:{bus:02x}:{dev:02x}.{fn}"
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text(f"{numa}\n")
    for n, cpus in nodes.items():
        d = root / f"sys/devices/system/node/node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")


def test_parse_cpulist():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity.parse_cpulist("") == []


def test_eight_gpus_two_sockets_get_disjoint_numa_local_shares(tmp_path):
    gpus = [(0, 0x05 + 0x10 * i, 0, 0, 0 if i < 4 else 1) for i in range(8)]
    _tree(tmp_path, gpus, {0: "0-47,96-143", 1: "48-95,144-191"})
    allowed = list(range(192))
    plan = affinity.plan(list(range(8)), allowed, root=str(tmp_path), env={})
    assert [p["numa_node"] for p in plan] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert plan[0]["pci"] == "0000:05:00.0" and plan[7]["pci"] == "0000:75:00.0"
    node0 = set(affinity.parse_cpulist("0-47,96-143"))
    seen = set()
    for r, p in enumerate(plan):
        cpus = set(p["cpus"])
        assert len(cpus) == 24 and not (cpus & seen)  # 96 cores of a node over its four ranks
        assert (cpus <= node0) == (r < 4)
        seen |= cpus
    assert seen == set(allowed)


def test_ranks_sharing_one_gpu_split_its_node(tmp_path):
    _tree(tmp_path, [(0, 0x75, 0, 0, 1)], {0: "0-31", 1: "32-63"})
    plan = affinity.plan([0, 0], list(range(64)), root=str(tmp_path), env={})
    assert plan[0]["cpus"] == list(range(32, 48)) and plan[1]["cpus"] == list(range(48, 64))


def test_cores_outside_the_jobs_mask_are_never_used(tmp_path):
    _tree(tmp_path, [(0, 0x75, 0, 0, 0)], {0: "0-63"})
    plan = affinity.plan([0], [4, 5, 6, 7], root=str(tmp_path), env={})
    assert plan[0]["cpus"] == [4, 5, 6, 7]


def test_without_numa_information_the_jobs_cores_are_split_evenly(tmp_path):
    _tree(tmp_path, [(0, 0x75, 0, 0, -1), (0, 0x76, 0, 0, -1)], {})
    plan = affinity.plan([0, 1], list(range(10)), root=str(tmp_path), env={})
    assert plan[0]["cpus"] == [0, 1, 2, 3, 4] and plan[1]["cpus"] == [5, 6, 7, 8, 9]
    assert plan[0]["numa_node"] is None and "no NUMA information" in plan[0]["how"]
    # no sysfs at all (not even the GPUs): the same fallback
    plan = affinity.plan([0, 1, 2], list(range(4)), root=str(tmp_path / "nowhere"), env={})
    assert [p["cpus"] for p in plan] == [[0, 1], [2], [3]]


def test_visible_devices_reorder_the_topology(tmp_path):
    _tree(tmp_path, [(0, 0x05, 0, 0, 0), (0, 0x15, 0, 0, 1)], {0: "0-3", 1: "4-7"})
    plan = affinity.plan([0], list(range(8)), root=str(tmp_path), env={"HIP_VISIBLE_DEVICES": "1"})
    assert plan[0]["pci"] == "0000:15:00.0" and plan[0]["cpus"] == [4, 5, 6, 7]


def test_pin_rank_sets_the_mask_and_reports_it(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    _tree(tmp_path, [(0, 0x75, 0, 0, 0)], {0: ",".join(str(c) for c in allowed)})
    try:
        me = affinity.pin_rank(1, [0, 0], root=str(tmp_path))
        assert me["pinned"] and sorted(os.sched_getaffinity(0)) == me["cpus"]
        assert set(me["cpus"]) <= set(allowed) and (len(allowed) < 2 or len(me["cpus"]) < len(allowed))
    finally:
        os.sched_setaffinity(0, allowed)


def test_hip_and_cuda_visible_devices_are_aliases_not_composed():
    """launchers set both to the same permutation: one re-indexing, HIP_VISIBLE_DEVICES first"""
    assert affinity.visible_ordinals(4, {"HIP_VISIBLE_DEVICES": "1,0", "CUDA_VISIBLE_DEVICES": "1,0"}) == [1, 0]
    assert affinity.visible_ordinals(4, {"CUDA_VISIBLE_DEVICES": "2,3"}) == [2, 3]
    assert affinity.visible_ordinals(4, {"HIP_VISIBLE_DEVICES": "3", "CUDA_VISIBLE_DEVICES": "0"}) == [3]
    # ROCR_VISIBLE_DEVICES re-indexes below HIP: the two do compose
    assert affinity.visible_ordinals(4, {"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "1"}) == [3]


def _siblings(root, pairs):
    for a, b in pairs:
        for c in (a, b):
            d = root / f"sys/devices/system/cpu/cpu{c}/topology"
            d.mkdir(parents=True)
            (d / "thread_siblings_list").write_text(f"{a},{b}\n")


def test_shares_are_dealt_by_physical_core(tmp_path):
    """a core and its SMT sibling go to the same rank (a contiguous split of "0-47,96-143" gave rank 0
    the cores 0-23 and rank 2 their siblings 96-119)"""
    gpus = [(0, 0x05 + 0x10 * i, 0, 0, 0) for i in range(4)]
    _tree(tmp_path, gpus, {0: "0-47,96-143"})
    _siblings(tmp_path, [(c, c + 96) for c in range(48)])
    plan = affinity.plan(list(range(4)), list(range(192)), root=str(tmp_path), env={})
    seen = set()
    for r, p in enumerate(plan):
        cpus = set(p["cpus"])
        assert len(cpus) == 24 and not (cpus & seen)
        assert cpus == {c for c in range(12 * r, 12 * r + 12)} | {c + 96 for c in range(12 * r, 12 * r + 12)}
        seen |= cpus
