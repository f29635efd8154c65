"""BVH build knobs against node visits / triangle tests / frame time on C4 (pool pipeline): tools/bvh_quality.py"""
import sys, os, tempfile, itertools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
configs = [dict(), dict(PRT_SAH_BINS="32"), dict(PRT_SAH_BINS="64"), dict(PRT_SAH_SWEEP="64"), dict(PRT_SAH_SWEEP="1024"),
           dict(PRT_SAH_BINS="32", PRT_SAH_SWEEP="256"), dict(PRT_SAH_SWEEP="16384")]
if len(sys.argv) > 1 and sys.argv[1] == "leaves":       # leaf size limit x SAH traversal cost
    configs = [dict(PRT_LEAF_MAX=str(l), PRT_SAH_TRAV_COST=t) for l in (4, 3, 2, 1) for t in ("1", "0.5", "2")] + [dict()]
for cfg in configs:
    for k in ("PRT_SAH_BINS", "PRT_SAH_SWEEP", "PRT_LEAF_MAX", "PRT_SAH_TRAV_COST"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    r = api.Renderer(0); info = r.upload(hs)
    p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    ms = min(r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True).render_ms for _ in range(4))
    c = r.render_device(cam, api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS), w, h, 0, w * h, buf.data_ptr(), True)
    print("%-44s build %.0f ms nodes %d depth %d | %.3f ms/frame | node visits %.3f/ray tri tests %.3f/ray rays %d" % (
        cfg, info.bvh_build_ms, info.bvh_node_count, info.bvh_max_depth, ms, c.node_visits / c.ray_count, c.tri_tests / c.ray_count, c.ray_count), flush=True)
    r.close()
