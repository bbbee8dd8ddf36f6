import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    """(arrays, meta) captured from the reference by tests/golden/make_golden.py."""
    arrays = np.load(os.path.join(GOLDEN, "reference_vectors.npz"))
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def golden_r02():
    """(arrays, meta) of tests/golden/make_golden_r02.py: real-size forwards, configs[0] / [4] loops, configs[2] cells."""
    arrays = np.load(os.path.join(GOLDEN, "reference_vectors_r02.npz"))
    with open(os.path.join(GOLDEN, "reference_vectors_r02.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def golden_r03():
    """(arrays, meta) of tests/golden/make_golden_r03.py: compute_trajectory_metrics_batch aggregates, the student-resize
    branches, one cell of the eight-scale grid of configs[3]."""
    arrays = np.load(os.path.join(GOLDEN, "reference_vectors_r03.npz"))
    with open(os.path.join(GOLDEN, "reference_vectors_r03.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def models():
    """Seeded synthetic models (CPU weight containers), keyed by size factor; digests checked against golden."""
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    cache = {}

    def get(sf):
        if sf not in cache:
            cfg = Config()
            cfg.image_size = 16
            cache[sf] = make_model(DiffusionUNet, cfg, sf)
        return cache[sf]
    return get
