import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle, pinned-math flavour (what the GPU is compared against bit for bit)."""
    from oracle.oracle import Oracle, build
    build()
    return Oracle("pinned")


@pytest.fixture(scope="session")
def oracle_libm():
    """glibc sincosf/powf flavour: reproduces the numbers SURVEY.md Appendix C recorded."""
    from oracle.oracle import Oracle, build
    build()
    return Oracle("libm")


@pytest.fixture(scope="session")
def bunny_matte():
    from rtcuda_amd import scenes
    return scenes.cornell_bunny("matte")


@pytest.fixture(scope="session")
def bunny_full_bsdf():
    from rtcuda_amd import scenes
    return scenes.cornell_bunny("full_bsdf")


def default_camera(oracle, aspect):
    return oracle.camera((0.5, 0.5, 1.5), (0.5, 0.5, 0.0), (0.0, 1.0, 0.0), 37.8, aspect)
