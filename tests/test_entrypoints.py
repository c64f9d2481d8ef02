"""CPU checks of the driver-facing entry points: they must import and parse without a GPU."""
import ast
import importlib
import os
import subprocess
import sys

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
