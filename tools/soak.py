"""Soak: N frames of C4 on the pool pipeline, every frame's pixels hashed - all frames must be the same bits (the image is a pure
function of scene, seed and pixel), ray counts equal; then the same for adaptive mode and a two-light scene."""
import sys, os, tempfile, hashlib, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
def soak(scene, lm, w, h, spp, max_spp, frames):
    s = scenes.make_scene(scene); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
    hs = api.HostScene(d, "scene.obj", lm, s.camera_position)
    r = api.Renderer(0); r.upload(hs)
    cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
    p = api.default_params(spp, 1234, max_spp=max_spp)
    buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    seen, rays, t0 = set(), set(), time.time()
    for k in range(frames):
        c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
        seen.add(hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()); rays.add(int(c.ray_count))
    print("%-14s light mode %d %dx%d spp %d..%d: %d frames in %.1f s, %d distinct images, %d distinct ray counts (%d rays)" % (
        scene, lm, w, h, spp, max_spp or spp, frames, time.time() - t0, len(seen), len(rays), max(rays)), flush=True)
    r.close()
    return len(seen) == 1 and len(rays) == 1
ok = soak("terrain_1m", 0, 1920, 1080, 8, 0, n)
ok &= soak("terrain_1m", 0, 960, 540, 10, 50, max(4, n // 10))
ok &= soak("many_materials", 1, 640, 360, 4, 0, n)
ok &= soak("coincident", 2, 640, 360, 4, 0, n)
print("soak ok" if ok else "SOAK FAILED")
sys.exit(0 if ok else 1)
