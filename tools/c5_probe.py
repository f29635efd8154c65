"""C5 (4K x 64 spp, depth 8) on the pool pipeline: timed frames, then one counting frame with the DEBUG_UTIL diagnostics.
    python tools/c5_probe.py [frames]"""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 3840, 2160
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
p = api.default_params(64, 1234, pipeline=capi.PIPELINE_POOL, bounce_depth=8)
r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
ms = []
for _ in range(frames):
    c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    ms.append(c.render_ms)
print("C5: %s ms, %d rays, %.1f Mrays/s at the median, %d launches" % (["%.1f" % m for m in ms], c.ray_count, c.ray_count / sorted(ms)[len(ms) // 2] / 1e3, c.trace_kernel_launches), flush=True)
r.set_option("DEBUG_UTIL", 1)
p.pipeline = capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS
c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
st = r.render_stats()
print("counting frame %.1f ms: node visits %d (%.1f per traced ray), triangle tests %d, shaded hits %d, elided shadow rays %d, parked %d + %d" % (
    c.render_ms, c.node_visits, c.node_visits / max(1, c.ray_count - st.elided_shadow_rays), c.tri_tests, c.shaded_hits, st.elided_shadow_rays, st.parked_rays, st.parked_shadow_rays), flush=True)
