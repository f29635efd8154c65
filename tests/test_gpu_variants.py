"""The option libraries in the driver's GPU suite.

`make hip-bvh8` builds libprt_hip_bvh8.so: the same ABI over the 8-wide compressed BVH (csrc/dev_trace8.h, 415 lines that the
default library does not contain).  The suite loads one library per process (PRT_HIP_LIB, par_raytracer_amd/capi.py), so the
variant runs in a CHILD process: three of the reference's golden fixtures - the Cornell box, the coincident-geometry scene
with two lights (near ties decided by the reference's visit order) and the headline 1M-triangle frame - on both production
pipelines, ray counts equal and RGB within 1e-4, as tests/test_gpu_parity.py asks of the default library.
__graft_entry__.build() builds the variant, so the driver's run covers it; where the file is missing the test is skipped.
"""
from __future__ import annotations

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np
from conftest import camera_and_params, host_scene, load_golden
from par_raytracer_amd import api, capi
lib = capi.hip_lib()
assert os.path.basename(lib._name) == %(lib)r, lib._name
assert lib.prt_abi_version() == 5
assert bool(lib.prt_build_flags() & capi.BUILD_BVH4) == %(bvh4)r
for name in %(fixtures)r:
    g = load_golden(name)
    r = api.Renderer(0)
    r.upload(host_scene(str(g["scene"]), int(g["light_mode"])))
    assert r.scene_info().bvh_node_bytes == %(node_bytes)d
    for pipeline in (2, 4):
        cam, p = camera_and_params(g, pipeline)
        img, ctr = r.render_lattice(cam, p, int(g["width"]), int(g["height"]), int(g["lattice"]))
        d = float(np.abs(img[:, :, :3] - g["rgb"]).max())
        assert ctr.ray_count == int(g["ray_count"]), (name, pipeline, ctr.ray_count, int(g["ray_count"]))
        assert d <= 1e-4, (name, pipeline, d)
        print("%%s pipeline %%d: rays equal, max |dRGB| %%.2e" %% (name, pipeline, d), flush=True)
    r.close()
print("variant ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("lib,bvh4,node_bytes", [("libprt_hip_bvh8.so", False, 80)])
def test_option_library_reproduces_the_reference_goldens(lib, bvh4, node_bytes):
    path = os.path.join(ROOT, "par_raytracer_amd", lib)
    if not os.path.exists(path):
        pytest.skip("%s not built (make hip-bvh8)" % lib)
    env = dict(os.environ)
    env["PRT_HIP_LIB"] = lib
    code = CHILD % {"root": ROOT, "lib": lib, "bvh4": bvh4, "node_bytes": node_bytes,
                    "fixtures": ["c2_cornell_128", "coincident_two_lights", "c4_terrain1m_1080p_l40"]}
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0 and b"variant ok" in out.stdout, (out.returncode, out.stdout.decode()[-1500:], out.stderr.decode()[-3000:])
