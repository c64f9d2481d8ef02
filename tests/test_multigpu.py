"""The multi-GPU story of this path -- contiguous stream blocks per rank, no collective on the data path,
control plane on gloo -- exercised for real on the one GPU of the test box:

* every rank is a fresh child process that owns its own handle on GPU 0 (a GPU is shared, the code path is the
  N-rank one: rendezvous, ``shard.stream_range``, per-rank handles and look-back state, ``to_global`` +
  ``gather_records``, barrier, max over ranks, rank-0 print);
* a stream analysed inside a rank's block must give the records it gives inside the single batch, byte for byte
  (the reference runs one analyzer per device, radiotracking/__main__.py:118-140).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pyradiotracking_amd import _native, shard, synth

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FS, NPERSEG, SEGS = 2048000, 256, 500  # per buffer: 500 segments, two consecutive buffers
N_STREAMS = 7  # not a multiple of 2 or 3: uneven blocks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _population():
    """[S, 2, B] complex64: two consecutive buffers per stream; pulses may straddle the boundary (look-back live)."""
    from pyradiotracking_amd.analyze import window_coefficients

    blen = NPERSEG * SEGS
    w = window_coefficients("hamming", NPERSEG)
    out = []
    for s in range(N_STREAMS):
        rng = np.random.default_rng([77, s])
        pulses = synth.random_pulses(rng, 2 * blen, FS, w, 9)
        # one pulse per stream placed across the buffer boundary on purpose
        pulses.append(synth.Pulse(blen - int(0.004 * FS) - 37 * s, int(0.012 * FS), 1e5 * (s - 3), synth.amp_for_peak_dbw(-70.0, w, FS)))
        out.append(synth.make_stream(synth.StreamSpec(2 * blen, FS, pulses), seed=500 + s).reshape(2, blen))
    return np.stack(out)


def _analyse(iq_block, first_stream, lanes=1):
    """records of two consecutive buffers of a block of streams, on a handle of its own"""
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer

    n = iq_block.shape[0]
    an = BatchSignalAnalyzer([str(first_stream + i) for i in range(n)], sdr_callback_length=iq_block.shape[2], sample_rate=FS,
                             fft_nperseg=NPERSEG, gpu=0, lanes=lanes)
    recs = []
    for k in range(2):
        an.enqueue(np.ascontiguousarray(iq_block[:, k]))
        recs.append(an.fetch_records())
    an.close()
    return recs


def _rank_main(rank, world, port, q):
    """one rank = one process: own block, own handle, gather over gloo"""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        iq = _population()
        lo, hi = shard.stream_range(rank, world, N_STREAMS)
        mine = _analyse(iq[lo:hi], lo)
        merged = [shard.gather_records(shard.to_global(r, rank, world, N_STREAMS)) for r in mine]
        if rank == 0:
            whole = _analyse(iq, 0)  # the single batch, same GPU, another handle
            q.put(dict(
                equal=[m.tobytes() == w.tobytes() for m, w in zip(merged, whole)],
                n=[len(w) for w in whole],
                negative_starts=int(sum(int((w["start"] < 0).sum()) for w in whole)),
                streams_with_records=[sorted(set(int(s) for s in w["stream"])) for w in whole],
            ))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_records_equal_single_batch(world):
    """stream i analysed on rank r == stream i analysed in the single batch, over two buffers (look-back included)"""
    if _native.device_count() < 1:
        pytest.fail("no GPU visible")
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = q.get(timeout=300)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got["equal"] == [True, True], got
    assert min(got["n"]) > 0
    assert got["negative_starts"] > 0, "no record reached back into the previous buffer: the look-back was not exercised"
    assert got["streams_with_records"][1] == list(range(N_STREAMS))


def test_blocks_on_separate_handles_equal_single_batch_in_process():
    """the same statement without processes: 2 and 3 blocks, one handle each, plus the two-lane handle"""
    iq = _population()
    whole = _analyse(iq, 0)
    for world in (2, 3):
        parts = [[], []]
        for r in range(world):
            lo, hi = shard.stream_range(r, world, N_STREAMS)
            for k, rec in enumerate(_analyse(iq[lo:hi], lo)):
                parts[k].append(shard.to_global(rec, r, world, N_STREAMS))
        for k in range(2):
            assert np.concatenate(parts[k]).tobytes() == whole[k].tobytes(), (world, k)
    laned = _analyse(iq, 0, lanes=2)
    assert [a.tobytes() for a in laned] == [a.tobytes() for a in whole]


def _run_bench(n_gpus, extra, launcher=True):
    env = dict(os.environ, RT_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", str(n_gpus), "--steps", "3", "--warmup", "1", "--settle", "2", "--isolated-steps", "2", "--cpu-streams", "4",
            "--parity-streams", "4"] + extra
    if n_gpus == 1 or not launcher:
        cmd = [sys.executable, os.path.join(REPO, "bench.py")] + args  # N > 1: bench.py starts its ranks itself
    else:
        # the driver's launch line; the launcher is a fresh process that never touches the GPU itself
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu_weak():
    """`bench.py --gpus 2` as the driver launches it (two ranks, here both on GPU 0): one JSON line, n_gpus 2"""
    d = _run_bench(2, ["--streams", "32"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["streams_total"] == 64
    assert d["value"] > 0 and d["config"]["fallbacks"] == 0 and d["config"]["records_per_step"] > 0
    assert d["parity"]["streams_checked"] == 4 and d["parity"]["streams_mismatched"] == 0  # first + last of both shards
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` typed by hand (no RANK / WORLD_SIZE): the script spawns the two ranks itself before it
    touches a GPU, relays rank 0's one JSON line, names each rank's device"""
    d = _run_bench(2, ["--streams", "32"], launcher=False)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["streams_total"] == 64
    assert d["value"] > 0 and d["config"]["fallbacks"] == 0 and d["config"]["records_per_step"] > 0
    assert d["parity"]["streams_checked"] == 4 and d["parity"]["streams_mismatched"] == 0
    devs = d["config"]["devices"]
    assert [x["rank"] for x in devs] == [0, 1] and all(x["ordinal"] == 0 for x in devs)  # RT_BENCH_SHARE_GPU: both on GPU 0
    assert all(x["name"] for x in devs)
    # every rank pinned itself before its first GPU call; two ranks on one GPU (one NUMA node) get disjoint core sets
    # (the reference: taskset per analyzer, __main__.py:122-128)
    from pyradiotracking_amd import affinity

    sets = [set(affinity.parse_cpulist(x["cpus"]["cpulist"])) for x in devs]
    assert all(x["cpus"]["pinned"] and x["cpus"]["n"] == len(c) > 0 for x, c in zip(devs, sets)), devs
    if len(os.sched_getaffinity(0)) >= 2:
        assert not (sets[0] & sets[1]), devs


def test_bench_fails_fast_when_a_rank_dies_before_the_rendezvous():
    """A rank that exits before it has joined (bad device ordinal, out of memory while generating IQ, an import error):
    `python bench.py --gpus 2` stops the other rank by its PID and returns non-zero with the failed rank named, within
    seconds -- not after the rendezvous store's time-out (reference: the supervisor loop that ends all analyzers when one
    dies, __main__.py:152-190)."""
    import time

    env = dict(os.environ, RT_BENCH_SHARE_GPU="1", RT_BENCH_FAIL_RANK="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--settle", "2", "--streams", "32"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    took = time.monotonic() - t0
    assert r.returncode != 0 and took < 60, (r.returncode, took, r.stderr[-2000:])
    assert "rank 1 failed (exit code 3)" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no result line of a run that did not happen


def test_bench_line_prices_the_isolated_launch():
    """`roofline.frac` / `kernel_ms` describe the scan launch alone (one lane); the per-launch figures of the three-lane
    timed region sit beside them; the host sinks are reported outside `value`"""
    d = _run_bench(1, ["--streams", "32"])
    r = d["roofline"]
    assert r["kernel_ms"] > 0 and r["kernel_ms_concurrent"] > 0 and r["launches_per_step_timed_region"] == 3
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"]) < 2e-3 * max(r["frac"], 1e-3) + 1e-4
    assert r["algorithmic_bytes_per_launch"] == 32 * 8000 * 256 * 8
    h = d["host_sinks"]
    assert h["records_per_s_produced"] > 0 and h["signal_objects_per_s"] > 0 and h["csv_rows_per_s"] > 0
    assert d["config"]["devices"][0]["ordinal"] == 0 and len(d["config"]["devices"]) == 1


def test_bench_strong_scaling_population_is_the_same_at_every_n():
    """config 4 geometry, one fixed population of 48 streams: sharded over 1, 2 and 3 ranks it yields the same
    number of records and candidate cells, and every rank's boundary streams match the oracle"""
    runs = {n: _run_bench(n, ["--workload", "config4", "--total-streams", "48"]) for n in (1, 2, 3)}
    for n, d in runs.items():
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["streams_total"] == 48
        assert d["config"]["workload"].startswith("config4")
        assert d["parity"]["streams_mismatched"] == 0 and d["config"]["fallbacks"] == 0
    assert len({d["config"]["records_per_step"] for d in runs.values()}) == 1, {n: d["config"]["records_per_step"] for n, d in runs.items()}
    assert len({d["config"]["candidate_cells_per_step"] for d in runs.values()}) == 1
    assert runs[1]["config"]["records_per_step"] > 0 and "cpu_baseline" in runs[1]


def test_bench_other_configs_block_is_timed_in_the_same_run(monkeypatch):
    """`other_configs`: after the headline the same process times further configurations (scaled down here), each with its
    own parity check against the oracle, kernel time and fractions; the headline keys stay what they were"""
    import bench

    small = [
        ("config3", dict(streams=24, sample_rate=2400000, samples=240000, nperseg=1024, window="hann", trains=False, lanes=1, what="scaled down")),
        ("default_geometry_noise_floor", dict(streams=32, sample_rate=300000, samples=300000, nperseg=256, window="hamming", trains=False, lanes=2,
                                              noise_dbw=-88.0, settle=6, what="scaled down")),
    ]
    env_cfg = json.dumps(small)
    env = dict(os.environ, RT_BENCH_OTHER_CONFIGS_JSON=env_cfg)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--settle", "2", "--isolated-steps", "2",
                        "--streams", "32", "--cpu-streams", "4", "--parity-streams", "4", "--other-configs", "on", "--other-steps", "3"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["workload"].startswith("config2") and d["value"] > 0
    oc = d["other_configs"]
    assert [o["name"] for o in oc] == ["config3", "default_geometry_noise_floor"], oc
    for o in oc:
        assert "failed" not in o and "skipped" not in o, o
        assert o["value"] > 0 and o["kernel_ms"] > 0 and 0 < o["frac"] < 1 and 0 < o["whole_path_frac"] < 1
        assert o["parity_streams_checked"] == 4 and o["parity_streams_mismatched"] == 0, o
        assert o["steps"] == 3
    assert oc[1]["mode"] in ("runfilter", "dense") and oc[0]["mode"] == "sparse", oc
    assert bench.OTHER_CONFIGS[1][1]["streams"] == 32768  # the real block: all of config 4 on one GPU


def test_bench_sharded_configs_block_carries_north_stars_curve():
    """At N > 1 the bare `bench.py --gpus N` times, after the weak-scaled headline, BASELINE configs 4 and 5 as ONE population
    each sharded over the same ranks (`sharded_configs`; populations scaled down here, two ranks sharing the test box's GPU):
    every rank's boundary streams match the oracle, the shards add up to the population, and the block at world 1 reports the
    same records -- a population is the same however it is sharded (the reference: one analyzer process per SDR,
    radiotracking/__main__.py:118-140)."""
    small = [
        ("config4", dict(total=48, sample_rate=2048000, samples=524288, nperseg=256, window="hamming", trains=False, what="scaled down")),
        ("config5", dict(total=10, sample_rate=3200000, samples=3200000, nperseg=4096, window="hamming", trains=True, what="scaled down")),
    ]
    os.environ["RT_BENCH_SHARDED_CONFIGS_JSON"] = json.dumps(small)
    try:
        runs = {n: _run_bench(n, ["--streams", "32", "--sharded-configs", "on", "--other-configs", "off", "--other-steps", "3"]) for n in (2, 1)}
    finally:
        del os.environ["RT_BENCH_SHARDED_CONFIGS_JSON"]
    for n, d in runs.items():
        assert d["n_gpus"] == n and d["scaling"] == "weak" and d["config"]["workload"].startswith("config2") and d["value"] > 0  # headline keys untouched
        sc = d["sharded_configs"]
        assert [c["name"] for c in sc] == ["config4", "config5"], sc
        for c, (_, spec) in zip(sc, small):
            assert "failed" not in c and "skipped" not in c, c
            assert c["scaling"] == "strong" and c["n_gpus"] == n and c["value"] > 0 and c["steps"] == 3
            assert len(c["per_rank_ms"]) == n and all(ms > 0 for ms in c["per_rank_ms"])
            assert sum(c["streams_per_rank"]) == spec["total"] and len(c["streams_per_rank"]) == n
            assert c["parity_streams_checked"] == 2 * n and c["parity_streams_mismatched"] == 0, c
            assert c["fallbacks"] == 0 and c["records_per_step"] > 0
            assert c["speedup_vs_n1_reference"] is None  # (the reference on file is of the full population)
    for a, b in zip(runs[1]["sharded_configs"], runs[2]["sharded_configs"]):
        assert a["records_per_step"] == b["records_per_step"] and a["candidate_cells_per_step"] == b["candidate_cells_per_step"]
    import bench

    assert [(n, sp["total"]) for n, sp in bench.SHARDED_CONFIGS] == [("config4", 32768), ("config5", 8192)]  # the real block: north_star's populations
