"""The device traversal code on the host: par_raytracer_amd/csrc/dev_trace*.h compiled with g++ against tests/hip_shim and run
one lane at a time (tests/trace_host_harness.cpp).  For thousands of rays over a scene with doubled and coplanar triangles the
hit trace_ray() returns - through the BVH the library builds and, where hits are near-tied, through resolve_near_ties() - must be
the hit of the reference's own sequential filter over every triangle in visit order (raytracer.cpp:104, 149, 208-220), bit for
bit; any-hit rays must agree on occluded / unoccluded.  All three traversal flavours the sources can be built as."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flavour", ["bvh8", "bvh8_octant", "bvh4"])
def test_device_traversal_equals_the_references_filter_on_the_host(flavour, tmp_path):
    exe = str(tmp_path / ("trace_host_" + flavour))
    flags = {"bvh4": [], "bvh8": ["-DPRT_BVH8"], "bvh8_octant": ["-DPRT_BVH8", "-DPRT_BVH8_OCTANT"]}[flavour]
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "tests", "hip_shim"),
           "-I" + os.path.join(ROOT, "par_raytracer_amd", "csrc")] + flags + [
           os.path.join(ROOT, "tests", "trace_host_harness.cpp"), os.path.join(ROOT, "par_raytracer_amd", "csrc", "bvh_build.cpp"), "-o", exe]
    build = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert build.returncode == 0, build.stdout.decode()
    run = subprocess.run([exe, "32", "8000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = run.stdout.decode()
    assert run.returncode == 0, out
    assert "closest-hit mismatches 0, any-hit mismatches 0, unresolved near ties 0" in out
    assert " 0 with a near tie" not in out, "the scene is meant to exercise resolve_near_ties: " + out
