"""pytest configuration: marker registration, TZ pin, repo on sys.path."""
import os
import sys
import time

os.environ["TZ"] = "UTC"  # golden vectors pin relative offsets with TZ=UTC
time.tzset()

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
