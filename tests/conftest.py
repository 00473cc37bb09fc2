"""pytest configuration: the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` runs the oracle / host-logic / ABI-surface tests on any CPU box;
``-m gpu`` runs the parity tests proper on a MI355X through the C-ABI library.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")

# The CPU oracle is OpenMP code run on tiny models: on a many-core GPU host the default thread count makes every
# parallel region a spin-wait festival (seconds per forward).  Results do not depend on the thread count.
os.environ.setdefault("OMP_NUM_THREADS", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) and the built HIP library")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C oracle (built on demand with gcc)."""
    from oracle import binding as ob
    if not os.path.exists(ob.ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return ob.load_oracle()


@pytest.fixture(scope="session")
def ref_strict():
    """The compiled, unmodified reference (only where oracle/_ref was built)."""
    from oracle import binding as ob
    lib = ob.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref not built on this machine (needs /root/reference)")
    return lib


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("models"))


@pytest.fixture(scope="session")
def gold_ops():
    return np.load(os.path.join(GOLD, "ops.npz"))


_model_cache = {}


def synth_model(model_dir, preset, quant, gs=0, seed=39):
    """Write (once per session) a seeded synthetic model file; returns (path, spec)."""
    from nano_amd import modelfile as mf
    key = (preset, quant, gs, seed)
    if key not in _model_cache:
        spec = mf.preset(preset, quant, group_size=gs)
        path = os.path.join(model_dir, f"{preset}-{quant}-{gs}-{seed}.bin")
        mf.write_model(path, spec, seed=seed)
        _model_cache[key] = (path, spec)
    return _model_cache[key]


# every end-to-end golden case (tools/make_golden.py E2E_CASES): (preset, quant, group size)
E2E_CASES = [("tiny-nano", "f32", 0), ("tiny-nano", "q80", 32), ("tiny-nano", "q4k", 0),
             ("tiny-nano-odd", "f32", 0), ("tiny-nano-odd", "q80", 32), ("tiny-nano-odd", "q4k", 0),
             ("tiny-qwen3", "f32", 0), ("tiny-qwen3", "q80", 64), ("tiny-qwen3", "q4k", 0),
             # loader branches: Qwen2 architecture, un-shared classifier, the Nano exporter's default group size
             ("tiny-qwen2", "f32", 0), ("tiny-qwen2", "q80", 32), ("tiny-qwen2", "q4k", 0),
             ("tiny-nano-ucls", "q80", 32), ("tiny-nano-ucls", "f32", 0), ("tiny-nano", "q80", 128)]


def e2e_golden(preset, quant, gs):
    """Path of the golden file of an end-to-end case."""
    tag = f"{preset}_{quant}" + (f"_gs{gs}" if quant == "q80" and gs == 128 and "nano" in preset else "")
    return os.path.join(GOLD, f"e2e_{tag}.npz")


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def rel_err(a, b):
    """max|a-b| / max|b| -- the north-star's logits metric."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
