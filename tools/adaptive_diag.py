"""Where does adaptive mode (10..50 spp, one sample per pixel in flight) lose against fixed spp?  C4, pool pipeline:
lane utilisation / phase split of counting renders, and plain timings for fixed 1 / 2 / 8 / 50 spp and adaptive shards."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
for mn, mx in ((10, 50), (1, 0), (2, 0), (8, 0), (50, 0)):
    p = api.default_params(mn, 1234, max_spp=mx, pipeline=capi.PIPELINE_POOL)
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    cs = [r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True) for _ in range(2)]
    ms = min(c.render_ms for c in cs)
    pc = api.default_params(mn, 1234, max_spp=mx, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS)
    c = r.render_device(cam, pc, w, h, 0, w * h, buf.data_ptr(), True)
    st = r.render_stats()
    ph = list(st.phase_cycles)
    print("spp %2d max %2d: %8.2f ms %10d rays %6.0f Mrays/s | nodes/ray %.1f tris/ray %.1f | util node %.1f%% tri %.1f%% | rays/refill %.1f | phases topup %.1f%% trace %.1f%% shade %.1f%% (finalise %.1f%%) | parked %d" % (
        mn, mx, ms, cs[0].ray_count, cs[0].ray_count / ms / 1e3, c.node_visits / c.ray_count, c.tri_tests / c.ray_count,
        100.0 * st.node_visits / (64.0 * st.wave_node_steps), 100.0 * st.tri_tests / (64.0 * st.wave_tri_steps),
        c.ray_count / max(1, st.wave_refills), 100.0 * ph[0] / ph[3], 100.0 * ph[1] / ph[3], 100.0 * ph[2] / ph[3], 100.0 * ph[4] / ph[3],
        st.parked_rays), flush=True)
p = api.default_params(10, 1234, max_spp=50)
for nr in (2, 4, 8):
    r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    c = r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    print("adaptive 10..50, shard 1/%d: %.1f ms, %d rays, %.0f Mrays/s" % (nr, c.render_ms, c.ray_count, c.ray_count / c.render_ms / 1e3), flush=True)
