"""pytest configuration: `gpu` marker + shared fixtures.

CPU tier  (-m "not gpu"): oracle vs golden vectors / live reference, host logic, C-ABI exports.
GPU tier  (-m gpu)      : parity of the HIP path (through the C-ABI) against the oracle + golden.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def expand_seed(label: str, index: int, seed: int) -> bytes:
    """SHAKE128(label || LE64(index) || LE64(seed))[:32] — same expander as oracle/gen_golden.py."""
    return hashlib.shake_128(label.encode() + index.to_bytes(8, "little") + seed.to_bytes(8, "little")).digest(32)


def seeds(label, n, seed):
    return np.frombuffer(b"".join(expand_seed(label, i, seed) for i in range(n)), np.uint8).reshape(n, 32).copy()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "mlkem_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_npz():
    return np.load(os.path.join(GOLDEN_DIR, "mlkem_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle.loader import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference build (oracle/_ref).  Present in the build container and, as a prebuilt
    .so, on the GPU box; tests that need it skip when it is absent."""
    from oracle.loader import Ref
    try:
        return Ref()
    except (FileNotFoundError, RuntimeError, OSError) as e:  # pragma: no cover
        pytest.skip(f"reference build unavailable: {e}")


def unhex(s):
    return np.frombuffer(bytes.fromhex(s), np.uint8).copy()


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
