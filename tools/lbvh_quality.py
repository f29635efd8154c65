"""The GPU LBVH builder's back ends against the host SAH tree on C4: upload + build time, nodes, frame time, node visits."""
import sys, os, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
configs = [("host SAH", dict()), ("lbvh hybrid, clusters <= 64", dict(PRT_BVH_BUILDER="lbvh")),
           ("lbvh hybrid, clusters <= 16", dict(PRT_BVH_BUILDER="lbvh", PRT_LBVH_CLUSTER="16")),
           ("lbvh hybrid, clusters <= 256", dict(PRT_BVH_BUILDER="lbvh", PRT_LBVH_CLUSTER="256")),
           ("lbvh hybrid, clusters <= 4096", dict(PRT_BVH_BUILDER="lbvh", PRT_LBVH_CLUSTER="4096")),
           ("lbvh plain radix tree", dict(PRT_BVH_BUILDER="lbvh", PRT_LBVH_PLAIN="1"))]
ref = None
for name, cfg in configs:
    for k in ("PRT_BVH_BUILDER", "PRT_LBVH_CLUSTER", "PRT_LBVH_PLAIN"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    r = api.Renderer(0)
    r.upload(hs)                                             # first upload of a context pays for one-off set-up
    t0 = time.perf_counter(); info = r.upload(hs); up = time.perf_counter() - t0
    p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    ms = min(r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True).render_ms for _ in range(4))
    c = r.render_device(cam, api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS), w, h, 0, w * h, buf.data_ptr(), True)
    img = buf.cpu().numpy().copy()
    if ref is None: ref = (img, c.ray_count)
    same = bool((img.view("uint32") == ref[0].view("uint32")).all()) and c.ray_count == ref[1]
    print("%-32s upload %.3f s (build %.0f ms) nodes %d depth %d | %.3f ms/frame | %.2f node visits %.2f tri tests per ray | same image: %s" % (
        name, up, info.bvh_build_ms, info.bvh_node_count, info.bvh_max_depth, ms, c.node_visits / c.ray_count, c.tri_tests / c.ray_count, same), flush=True)
    r.close()
