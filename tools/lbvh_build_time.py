"""Build-time split of the BVH builders on C4 (PRT_DEBUG_UTIL prints of bvh_build.cpp / prt_api.hip)."""
import sys, os, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PRT_DEBUG_UTIL"] = "1"
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
for name, cfg in (("host SAH", dict()), ("lbvh hybrid 64", dict(PRT_BVH_BUILDER="lbvh")), ("lbvh hybrid 16", dict(PRT_BVH_BUILDER="lbvh", PRT_LBVH_CLUSTER="16")),
                  ("lbvh plain", dict(PRT_BVH_BUILDER="lbvh", PRT_LBVH_PLAIN="1"))):
    for k in ("PRT_BVH_BUILDER", "PRT_LBVH_CLUSTER", "PRT_LBVH_PLAIN"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    r = api.Renderer(0)
    r.upload(hs)
    sys.stderr.write("== %s (second upload)\n" % name); sys.stderr.flush()
    t0 = time.perf_counter(); info = r.upload(hs); up = time.perf_counter() - t0
    sys.stderr.write("   upload %.3f s, bvh_build_ms %.1f\n" % (up, info.bvh_build_ms)); sys.stderr.flush()
    r.close()
