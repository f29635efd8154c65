"""The round-free pool kernel (POOL_FLOW=1, kernels_flow.h) against the round-based one, ONE process: small scenes first (images must
be identical bit for bit, ray counts equal), then the C4 frame, its shards and deep bounces.   python tools/flow_probe.py [small|big ...]"""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
import texture_fixtures  # noqa: F401
what = sys.argv[1:] or ["small", "big"]
def setup(name, lm=0):
    s = scenes.make_scene(name); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
    hs = api.HostScene(d, "scene.obj", lm, s.camera_position)
    r = api.Renderer(0); r.upload(hs)
    return s, r
ok = True
if "small" in what:
    for name, lm, w, h, spp, depth, rs, ss in (("cornell_box", 0, 128, 128, 4, 2, 1, 1), ("terrain_64", 0, 160, 90, 2, 3, 1, 1), ("terrain_64", 1, 37, 23, 3, 2, 1, 1),
                                                 ("coincident", 2, 96, 72, 2, 2, 1, 1), ("cornell_box", 2, 48, 48, 2, 5, 2, 2), ("textured_gallery", 0, 160, 120, 4, 2, 1, 1),
                                                 ("icosphere_l3", 1, 64, 48, 2, 8, 1, 1), ("many_materials", 1, 96, 72, 2, 3, 1, 1)):
        s, r = setup(name, lm)
        cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
        p = api.default_params(spp, 77, bounce_depth=depth, reflection_samples=rs, spec_samples=ss, pipeline=capi.PIPELINE_POOL)
        r.set_option("POOL_FLOW", 0)
        a, ca = r.render(cam, p, w, h)
        r.set_option("POOL_FLOW", 1)
        try:
            b, cb = r.render(cam, p, w, h)
            b2, cb2 = r.render(cam, p, w, h)
            same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), b2.view(np.uint32))
            print("%-18s lights %d %dx%d spp %d depth %d: rays %d / %d, shaded %d / %d, image %s (max |d| %.3g)" % (name, lm, w, h, spp, depth, ca.ray_count, cb.ray_count,
                  ca.shaded_hits, cb.shaded_hits, "IDENTICAL" if same else "DIFFERS", float(np.abs(a - b).max())), flush=True)
            ok = ok and same and ca.ray_count == cb.ray_count
        except RuntimeError as e:
            print("%-18s FAILED: %s" % (name, e), flush=True)
            ok = False
        r.close()
    print("small scenes:", "all identical" if ok else "MISMATCH", flush=True)
if "big" in what and ok:
    s, r = setup("terrain_1m")
    w, h = 1920, 1080
    cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
    buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    def compare(tag, p, nr, n):
        ref = None
        rows = r.shard_rows(h, 8, 0, nr)
        for name, fl, sh in (("rounds, shared", 0, 1), ("rounds, private", 0, 0), ("no rounds", 1, 0), ("rounds, shared", 0, 1), ("no rounds", 1, 0)):
            r.set_option("POOL_FLOW", fl); r.set_option("POOL_SHARED", sh)
            r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
            res = [r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True) for _ in range(n)]
            ms = sorted(c.render_ms for c in res)
            torch.cuda.synchronize()
            img = buf.reshape(-1, 4)[:rows * w].cpu().numpy().view(np.uint32).copy()
            if ref is None: ref = (img, res[0].ray_count)
            same = "identical" if np.array_equal(ref[0], img) and ref[1] == res[0].ray_count else "DIFFERS (%d words, rays %d vs %d)" % (int((ref[0] != img).sum()), ref[1], res[0].ray_count)
            print("%-22s %-16s min %8.3f  median %8.3f ms   image + rays %s" % (tag, name, ms[0], ms[len(ms) // 2], same), flush=True)
    pf = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
    for nr in (1, 2, 4, 8, 16): compare("C4 1/%d" % nr, pf, nr, 6)
    compare("C4 8 spp depth 8", api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL, bounce_depth=8), 1, 3)
