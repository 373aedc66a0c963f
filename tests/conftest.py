import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MESHES = os.path.join(ROOT, "pathtracer_amd", "assets", "meshes.bin")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def bundled(ob):
    """The reference's bundled scene, Game.InitializeScene(11), built by the oracle."""
    s = ob.Scene(MESHES)
    s.initialize_scene(11)
    s.build()
    return s


@pytest.fixture(scope="session")
def soup50k(ob):
    s = ob.Scene("")
    s.soup(50000)
    s.build()
    return s


@pytest.fixture(scope="session")
def cornell(ob):
    s = ob.Scene(MESHES)
    s.add_mesh("cornell")
    s.build()
    return s


def bundled_view(scene, width, height, frames=1):
    """Camera of Game.Initialize: Set((0,2,15), (0,3,0)) before the scene exists -> focal 20."""
    fs, _ = scene.camera((0, 2, 15), (0, 3, 0), width, height, use_scene=False, frames=frames, samples=frames)
    return fs


def rel_rmse(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(1e-12, np.sqrt(np.mean(b ** 2))))
