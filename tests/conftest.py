import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def cksum_str(arr) -> str:
    return bytes(np.asarray(arr).tolist()).decode()


# ---- observed parity margins --------------------------------------------------------------------------
# Every tolerance helper of the GPU parity tests records what it OBSERVED next to the bound it enforced; the session writes
# them to gpurun_out/parity_margins.json (copied to profiles/ per round) and prints the tightest margins, so that a bound
# far above the result it passes is visible instead of silent.
PARITY_MARGINS = []


def record_margin(what, observed_max, observed_mean, bound_max, bound_mean=None):
    PARITY_MARGINS.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what,
                           "max": float(observed_max), "mean": float(observed_mean), "bound_max": float(bound_max),
                           "bound_mean": None if bound_mean is None else float(bound_mean)})


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump(PARITY_MARGINS, f, indent=0)
    except OSError:
        pass
    worst = sorted(PARITY_MARGINS, key=lambda m: -(m["max"] / m["bound_max"] if m["bound_max"] else 0))[:8]
    print("\n[parity margins] observed max / bound (largest ratios of %d checks):" % len(PARITY_MARGINS))
    for m in worst:
        print(f"  {m['max']:.3e} / {m['bound_max']:.3e} = {m['max'] / m['bound_max']:.2f}   {m['test']} {m['what']}")
