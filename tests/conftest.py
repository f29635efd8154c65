"""Shared test plumbing.

Markers: ``gpu`` = needs a real MI355X (the parity tests proper, all through the C ABI).  Everything else
runs on CPU: oracle vs the reference's golden vectors, host logic, library symbol checks, gloo sharding.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu)")
    config.addinivalue_line("markers", "slow: large CPU-side scenes (still run by default)")


def _ensure_built():
    """Build the CPU-side libraries if they are missing (the HIP library cross-compiles without a GPU)."""
    pkg = os.path.join(ROOT, "par_raytracer_amd")
    need = [os.path.join(pkg, "libprt_hip.so"), os.path.join(pkg, "libprt_host.so"),
            os.path.join(ROOT, "oracle", "libprt_oracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()


_ensure_built()

_SCENE_DIRS = {}


def scene_dir(name: str):
    """(ObjScene, directory) with <directory>/scene.obj written once per session."""
    from par_raytracer_amd import scenes
    import texture_fixtures  # noqa: F401  (registers the textured gallery scenes)
    if name not in _SCENE_DIRS:
        d = tempfile.mkdtemp(prefix="prt_test_%s_" % name)
        s = scenes.make_scene(name)
        scenes.write_obj(s, d, "scene.obj")
        _SCENE_DIRS[name] = (s, d)
    return _SCENE_DIRS[name]


_HOST_SCENES = {}


def host_scene(name: str, light_mode: int = 0):
    from par_raytracer_amd import api
    key = (name, light_mode)
    if key not in _HOST_SCENES:
        s, d = scene_dir(name)
        _HOST_SCENES[key] = api.HostScene(d, "scene.obj", light_mode, s.camera_position)
    return _HOST_SCENES[key]


def load_golden(name: str):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("golden fixture %s not generated" % name)
    return np.load(path, allow_pickle=False)


def camera_and_params(g, pipeline: int = 0):
    """PrtCamera + PrtParams for a golden fixture."""
    from par_raytracer_amd import api
    cam = api.make_camera(float(g["fov"]), int(g["width"]), int(g["height"]), g["camera_position"], g["camera_facing"])
    p = api.default_params(int(g["spp"]), int(g["seed"]), bounce_depth=int(g["bounce_depth"]),
                           reflection_samples=int(g["reflection_samples"]), spec_samples=int(g["spec_samples"]),
                           pipeline=pipeline, max_spp=int(g["max_spp"]) if "max_spp" in g else 0)
    return cam, p


@pytest.fixture(scope="session")
def gpu_renderer_factory():
    """Renderer(0) per uploaded scene, cached for the session (GPU tests run in one process)."""
    from par_raytracer_amd import api
    cache = {}

    def get(name: str, light_mode: int = 0):
        key = (name, light_mode)
        if key not in cache:
            r = api.Renderer(0)
            r.upload(host_scene(name, light_mode))
            cache[key] = r
        return cache[key]

    yield get
    for r in cache.values():
        r.close()
