import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def weights():
    from oracle import bp_oracle as O

    return O.load_weights()


@pytest.fixture(scope="session")
def clip_22k():
    """tests/golden/vocadito_10.wav decoded + resampled by the product's own ingest."""
    from basic_pitch_amd import audio

    x, sr = audio.load(os.path.join(GOLDEN, "vocadito_10.wav"))
    assert sr == 22050
    return x


def make_windows(kind: str, n: int, seed: int = 0) -> np.ndarray:
    """Seeded synthetic windows (n, 43844) float32 — BASELINE.md §4 input definitions."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, (n, 43844)).astype(np.float32)
    if kind == "normal":
        return (rng.standard_normal((n, 43844)) * 0.01).astype(np.float32)
    if kind == "tones":
        t = np.arange(43844) / 22050.0
        out = []
        for i in range(n):
            f0 = 110.0 * 2 ** (rng.integers(0, 36) / 12.0)
            x = sum(rng.uniform(0.1, 0.4) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in (1, 2, 3))
            out.append(x + 1e-3 * rng.standard_normal(43844))
        return np.asarray(out, dtype=np.float32)
    raise ValueError(kind)
