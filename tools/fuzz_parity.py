"""Random configurations (scene, lights, depth, bounce samples, spp, image size, seed, adaptive bounds, pipeline) against the
CPU oracle: equal ray counts, RGB within 1e-4.  usage: fuzz_parity.py [n_configs] [rng_seed] [deep|shallow] [opts]
With `opts` every configuration also draws up to four scheduling / tuning options at random values (prt_set_option): none of them
may change a pixel or the ray count - or keep a kernel from ending (the setting is printed BEFORE the render)."""
import sys, os, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as orc
sys.path.insert(0, os.path.join(ROOT, "tests"))
from par_raytracer_amd import api, scenes
import texture_fixtures  # registers the textured gallery scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
names = ["coincident", "cornell_box", "sphere_plane", "icosphere_l3", "terrain_64", "many_materials", "textured_gallery", "terrain_192", "jpeg_gallery", "png_gallery", "bmp_gallery", "tga_gallery"]
dirs, hosts, rends = {}, {}, {}
bad = 0
t0 = time.time()
fuzz_opts = "opts" in sys.argv[3:]
# option -> values (prt_options.h clamps what is out of range; every one of these is a legal request)
OPTS = {"POOL_SHARED": [0, 1], "POOL_SHARED_CAP": [64, 128, 512, 1024], "POOL_GUIDED": [0, 1, 4, 16, 64], "POOL_GUIDED_MIN": [1, 3, 8, 64, 512],
        "POOL_FAIR": [1, 3, 4, 8, 16], "POOL_CAP": [64, 128, 192, 1024, 4096], "POOL_TOPUP": [1, 7, 64, 4096], "POOL_BLOCKS_PER_CU": [1, 2, 8],
        "KEEP_MIN": [1, 17, 40, 64], "NODE_MIN": [0, 1, 32, 64], "NODE_FRAC": [0, 1, 4, 8], "WORK_REVERSE": [0, 1], "WORK_SCATTER": [0, 1], "NO_TILES": [0, 1],
        "STACK_CAP": [2, 5, 24], "POOL_PARK_CAP": [8, 100000], "PASS_SAMPLES": [97, 4096, 100000], "CHAINS": [1, 2, 4], "SHADE_BLOCK": [256, 1024],
        "TRACE_BLOCKS_PER_CU": [1, 3, 8], "CHUNK_MIN": [64, 512], "POOL_MAX_SAMPLES": [0, 1000, 10000000], "TRACE_DEAD_SHADOW_RAYS": [0, 1]}
for i in range(n):
    sc = names[int(rng.integers(0, len(names)))]
    lm = int(rng.integers(0, 3))
    if sc not in dirs:
        s = scenes.make_scene(sc); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj"); dirs[sc] = (s, d)
    s, d = dirs[sc]
    if (sc, lm) not in hosts:
        hosts[(sc, lm)] = api.HostScene(d, "scene.obj", lm, s.camera_position)
        r = api.Renderer(0); r.upload(hosts[(sc, lm)]); rends[(sc, lm)] = r
    hs, r = hosts[(sc, lm)], rends[(sc, lm)]
    w, h = int(rng.integers(9, 90)), int(rng.integers(7, 70))
    deep = len(sys.argv) > 3 and sys.argv[3] == "deep"            # bounce depth up to the ABI's maximum of 16
    depth, rs, ss, spp = int(rng.integers(0, 17 if deep else 8)), int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(1, 7))
    if rs + ss >= 4 and depth > 5: depth = 5                      # keep the oracle's tree finite in wall-clock terms
    if depth > 7 and rs + ss > 2: rs, ss = 1, 1
    seed = int(rng.integers(0, 2 ** 63))
    adaptive = rng.integers(0, 4) == 0
    max_spp = spp + int(rng.integers(1, 9)) if adaptive else 0
    thr = float(rng.choice([0.0, 0.001, 0.05, 1.0])) if adaptive else 0.0
    cam = api.make_camera(s.fov * float(rng.uniform(0.6, 1.3)), w, h, s.camera_position, s.camera_facing)
    def params(pl):
        return api.default_params(spp, seed, bounce_depth=depth, reflection_samples=rs, spec_samples=ss, pipeline=pl, max_spp=max_spp, variance_threshold=thr)
    ref, c_ref = orc.render(hs.desc, cam, params(0), w, h, 1, 16)
    chosen = {}
    if fuzz_opts:
        for k in rng.choice(sorted(OPTS), size=int(rng.integers(0, 5)), replace=False):
            chosen[str(k)] = int(rng.choice(OPTS[str(k)]))
        print("cfg %d: %s lm %d %dx%d depth %d rs %d ss %d spp %d max_spp %d thr %g seed %d opts %r" % (i, sc, lm, w, h, depth, rs, ss, spp, max_spp, thr, seed, chosen), flush=True)
        for k, v in chosen.items(): r.set_option(k, v)
    for pl in ([4] if adaptive else [2, 4]):
        img, c = r.render(cam, params(pl), w, h)
        dmax = float(np.abs(img.reshape(h, w, 4)[:, :, :3] - ref[:, :, :3]).max())
        ok = c.ray_count == c_ref.ray_count and dmax <= 1e-4
        if not ok:
            bad += 1
            print("MISMATCH cfg %d: %s lm %d %dx%d depth %d rs %d ss %d spp %d max_spp %d thr %g seed %d pipeline %d opts %r: rays %d vs %d, max|d| %g" % (
                i, sc, lm, w, h, depth, rs, ss, spp, max_spp, thr, seed, pl, chosen, c.ray_count, c_ref.ray_count, dmax), flush=True)
    for k in chosen: r.set_option(k, None)
    if i % 20 == 19: print("%d configs, %d mismatches, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
print("done: %d configs, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
