"""WORK_REVERSE (pixels handed out last-to-first: the frame's tail is the top of the image) against the default order, one
process: C4 frame, shards, adaptive mode; images must be identical.   python tools/reverse_probe.py"""
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
def run(p, nr, n):
    r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    ms = sorted(r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True).render_ms for _ in range(n))
    torch.cuda.synchronize()
    rows = r.shard_rows(h, 8, 0, nr)
    return ms[0], ms[len(ms) // 2], buf.reshape(-1, 4)[:rows * w].cpu().numpy().view(np.uint32).copy()
for tag, p, n in (("fixed 8 spp", api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL), 8), ("adaptive 10..50", api.default_params(10, 1234, pipeline=capi.PIPELINE_POOL, max_spp=50), 2)):
    for nr in (1, 2, 4, 8, 16):
        res = {}
        for rnd in range(2):
            for rev in (0, 1):
                r.set_option("WORK_REVERSE", rev)
                res.setdefault(rev, []).append(run(p, nr, n))
        same = all(np.array_equal(res[0][0][2], x[2]) for v in res.values() for x in v)
        print("%-16s 1/%-2d  forward min %.3f %.3f  reversed min %.3f %.3f ms   images identical: %s" % (
            tag, nr, res[0][0][0], res[0][1][0], res[1][0][0], res[1][1][0], same), flush=True)
