"""CPU checks of the driver-facing entry points: they must import and parse without a GPU."""
import ast
import importlib
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_and_tools_parse():
    for rel in ("bench.py", "__graft_entry__.py", "tools/profile_traffic.py", "tools/pmc_summary.py", "tests/perf/latency_single_stream.py"):
        ast.parse(open(os.path.join(REPO, rel)).read(), filename=rel)


def test_graft_entry_exposes_build_and_smoke():
    sys.path.insert(0, REPO)
    mod = importlib.import_module("__graft_entry__")
    assert callable(mod.build) and callable(mod.smoke)


def test_bench_refuses_to_run_without_gpu():
    """The product path has no CPU fallback: on a GPU-less host bench.py exits with a message."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke()/build() and the cpu_baseline leg of bench.py may touch oracle/:
    the package and the tools must not."""
    for sub in ("pyradiotracking_amd", "tools", "include"):
        for root, _, files in os.walk(os.path.join(REPO, sub)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                    text = open(os.path.join(root, f)).read()
                    assert "import oracle" not in text and "from oracle" not in text, f
    bench = open(os.path.join(REPO, "bench.py")).read()
    at = bench.index("from oracle")
    assert bench.count("from oracle") == 1 and "import oracle" not in bench
    assert bench[:at].rsplit("\ndef ", 1)[-1].startswith("cpu_baseline(")


def _bench_module():
    sys.path.insert(0, REPO)
    return importlib.import_module("bench")


def test_bench_workloads_follow_baseline_configs():
    """--workload picks the BASELINE.json geometry; config 2 is weak-scaled, the others shard one population."""
    import argparse

    bench = _bench_module()

    def ns(**kw):
        base = dict(workload="config2", streams=None, total_streams=None, sample_rate=None, seconds=None, nperseg=None, window=None, trains=None)
        base.update(kw)
        return argparse.Namespace(**base)

    w = bench.resolve_workload(ns(), 1)
    assert (w["name"], w["scaling"], w["streams"], w["samples"], w["nperseg"]) == ("config2", "weak", 256, 2048000, 256)
    w = bench.resolve_workload(ns(workload="config4"), 8)
    assert (w["name"], w["scaling"], w["total"], w["samples"], w["nperseg"]) == ("config4", "strong", 32768, 524288, 256)
    w = bench.resolve_workload(ns(workload="config5"), 8)
    assert (w["name"], w["scaling"], w["total"], w["nperseg"], w["trains"]) == ("config5", "strong", 8192, 4096, True)
    w = bench.resolve_workload(ns(workload="config3"), 1)
    assert (w["name"], w["total"], w["window"], w["sample_rate"]) == ("config3", 4096, "hann", 2400000)
    # the geometry flags of tools/run_configs.sh name the same configurations (per-GPU stream counts: weak)
    w = bench.resolve_workload(ns(streams=4096, sample_rate=2400000, nperseg=1024, window="hann"), 1)
    assert (w["name"], w["scaling"], w["streams"], w["samples"]) == ("config3", "weak", 4096, 2400000)
    w = bench.resolve_workload(ns(streams=32768, sample_rate=2048000, seconds=0.256), 1)
    assert (w["name"], w["samples"]) == ("config4", 524288)
    w = bench.resolve_workload(ns(nperseg=512), 1)
    assert w["name"] == "custom"
    # lanes per GPU when the caller does not say: three up to nperseg 512 while a rank holds fewer than 16 384 streams (config 4: all
    # 32 768 on one GPU -> one lane, an eighth of them -> three), one where the scans are persistent grids
    for workload, world, want in (("config2", 1, 3), ("config4", 1, 1), ("config4", 4, 3), ("config3", 1, 1), ("config5", 1, 1)):
        a = ns(workload=workload, lanes=None)
        bench.resolve_workload(a, world)
        assert a.lanes == want, (workload, world, a.lanes)
    a = ns(workload="config3", lanes=2)
    bench.resolve_workload(a, 1)
    assert a.lanes == 2


def test_pmc_traffic_is_only_quoted_for_the_kernel_it_was_measured_on(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json carries the sha256 of the scan kernel's machine code (the symbol's bytes in the gfx950 code
    object of the built library): the figure is quoted while the library that runs holds that very kernel -- host-side edits
    and other kernels do not invalidate it, a different scan kernel does"""
    import json

    bench = _bench_module()
    have = bench.scan_kernel_sha256()
    assert have is not None and len(have) == 64  # (build() has run: the library is there)
    assert bench.scan_kernel_sha256(symbol="_ZN2rt9stft_scanILi1ELi1ELb0ELb1ELi0EEEvNS_10StftParamsE") not in (None, have)  # another instantiation, another hash
    assert bench.scan_kernel_sha256(symbol="no_such_kernel") is None
    f = tmp_path / "pmc_traffic.json"
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(f))
    assert bench.pmc_traffic(True, 2)[0] is None  # no file
    f.write_text(json.dumps({"scan_kernel_sha256": "0" * 64, "bytes_per_launch_256_streams": 4456600000}))
    assert bench.pmc_traffic(True, 2)[0] is None  # stale
    f.write_text(json.dumps({"scan_kernel_sha256": have, "bytes_per_launch_256_streams": 4456600000}))
    assert bench.pmc_traffic(True, 2)[0] == 2228300000
    assert bench.pmc_traffic(False, 2)[0] is None  # another workload
    # ... and host-side launch geometry is part of the key (advisor, round 4): another chunk length, another byte count
    f.write_text(json.dumps({"scan_kernel_sha256": have, "bytes_per_launch_256_streams": 4456600000, "segs_per_chunk": 32}))
    assert bench.pmc_traffic(True, 2, 32)[0] == 2228300000
    got, why = bench.pmc_traffic(True, 2, 25)
    assert got is None and "chunk length" in why


def test_bench_refuses_more_gpus_than_the_box_has_within_seconds():
    """`python bench.py --gpus N` asks a child process how many GPUs there are before it starts any rank (the parent stays
    GPU-free) and says so instead of leaving ranks to wait for one another; here: none."""
    import subprocess
    import time

    from pyradiotracking_amd import _native

    if _native.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "RT_BENCH_SHARE_GPU")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env, cwd=REPO)
    assert r.returncode != 0 and time.monotonic() - t0 < 60
    assert "GPU(s)" in r.stderr and not r.stdout.strip()


def test_product_library_has_no_laboratory_hooks():
    """The shipped library never consults the environment and carries none of the laboratory's switch names (VERDICT round 4,
    item 7): csrc/rt_diag.h turns RT_DIAG_ENV into a null pointer unless the build has -DRT_DIAG, and refuses -D switches of
    the laboratory in a product build."""
    import subprocess

    from pyradiotracking_amd import build

    with open(build.LIB, "rb") as f:
        blob = f.read()
    for name in (b"RT_TEST_FAIL_LANE", b"RT_EXP_", b"RT_STAMPS", b"RT_ABLATE"):
        assert name not in blob, name
    nm = subprocess.run(["nm", "-D", build.LIB], capture_output=True, text=True)
    assert nm.returncode == 0 and "getenv" not in nm.stdout
    hdr = os.path.join(build.CSRC, "rt_diag.h")
    ok = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", hdr], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    for switch in ("-DRT_STAMPS", "-DRT_ABLATE=3", "-DRT_EXP_NOBAR1", "-DRT_WAVE64_4096=0", "-DRT_PK_R3_MASK=0"):
        bad = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", switch, hdr], capture_output=True, text=True)
        assert bad.returncode != 0 and "laboratory switch" in bad.stderr, switch
        lab = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-DRT_DIAG", switch, hdr], capture_output=True, text=True)
        assert lab.returncode == 0, lab.stderr
