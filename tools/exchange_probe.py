"""Block-shared pools with the end-of-round ray exchange (POOL_EXCHANGE = rays a wave hands over, 0 = off), ONE process, ONE
box, alternating: the C4 frame, its shards, the adaptive 10..50 spp mode; every image of one setting must equal the first
setting's bit for bit.  Lane-slot and drain counters (DEBUG_UTIL) at the end.

    XMODES=0,16,24,32 python tools/exchange_probe.py [rounds [what ...]]      what: frame shards adaptive deep c5 counters
    XOPT=POOL_GUIDED XMODES=0,4,8,16 XFIXED=POOL_GUIDED_MIN=8 python tools/exchange_probe.py 2 adaptive      (another option, same harness)
"""
import sys, os, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
what = sys.argv[2:] or ["frame", "shards", "adaptive", "counters"]
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
MODES = [("x%d" % int(v), int(v)) for v in os.environ.get("XMODES", "0,16,24,32").split(",")]
XOPT = os.environ.get("XOPT", "POOL_EXCHANGE")          # (any integer option can be swept the same way)
if XOPT == "POOL_EXCHANGE" or os.environ.get("XSHARED"): r.set_option("POOL_SHARED", int(os.environ.get("XSHARED", "1")))
for kv in os.environ.get("XFIXED", "").split(","):
    if kv: r.set_option(kv.split("=")[0], int(kv.split("=")[1]))


def timed(fn, n):
    fn()
    ms = []
    for _ in range(n):
        c = fn()
        ms.append(c.render_ms)
    ms.sort()
    return ms[0], ms[len(ms) // 2], c


def compare(tag, fn, n, pixels):
    """fn() renders into buf; returns nothing.  Both modes, `rounds` times alternating; images compared."""
    ref = None
    for rd in range(rounds):
        for name, v in MODES:
            r.set_option(XOPT, v)
            buf.zero_()                                        # a pixel this mode does not render must not pass for rendered
            lo, med, c = timed(fn, n)
            torch.cuda.synchronize()
            img = buf.reshape(-1, 4)[:pixels].cpu().numpy().view(np.uint32)
            same = ""
            if ref is None:
                ref = (img.copy(), c.ray_count)
            else:
                same = "  image %s, rays %s" % ("IDENTICAL" if np.array_equal(ref[0], img) else "DIFFERS (%d words)" % int((ref[0] != img).sum()),
                                                 "equal" if ref[1] == c.ray_count else "DIFFER %d vs %d" % (ref[1], c.ray_count))
            print("%-28s %-8s min %8.3f  median %8.3f ms  %7.1f Mrays/s%s" % (tag, name, lo, med, c.ray_count / med / 1e3, same), flush=True)


if "frame" in what:
    p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
    compare("C4 full frame", lambda: r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True), 6, w * h)
if "shards" in what:
    p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
    for n in (2, 4, 8, 16):
        rows = r.shard_rows(h, 8, 0, n)
        compare("C4 shard 1/%d" % n, lambda: r.render_shard_device(cam, p, w, h, 8, 0, n, buf.data_ptr(), True), 8, rows * w)
if "adaptive" in what:
    p = api.default_params(10, 1234, pipeline=capi.PIPELINE_POOL, max_spp=50)
    compare("C4 adaptive 10..50", lambda: r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True), 2, w * h)
    rows = r.shard_rows(h, 8, 0, 8)
    compare("C4 adaptive 10..50, 1/8", lambda: r.render_shard_device(cam, p, w, h, 8, 0, 8, buf.data_ptr(), True), 3, rows * w)
if "deep" in what:
    p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL, bounce_depth=8)
    compare("C4 8 spp depth 8", lambda: r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True), 3, w * h)
if "c5" in what:
    w5, h5 = 3840, 2160
    cam5 = api.make_camera(s.fov, w5, h5, s.camera_position, s.camera_facing)
    buf5 = torch.zeros((h5, w5, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    p = api.default_params(64, 1234, pipeline=capi.PIPELINE_POOL, bounce_depth=8)
    keep = buf
    buf = buf5
    compare("C5 4K x 64 spp depth 8", lambda: r.render_device(cam5, p, w5, h5, 0, w5 * h5, buf5.data_ptr(), True), 2, w5 * h5)
    buf = keep
if "counters" in what:
    r.set_option("DEBUG_UTIL", 1)
    for name, v in MODES:
        r.set_option(XOPT, v)
        for nr in (1, 8):
            p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS)
            sys.stderr.write("== exchange %s, fixed 8 spp, nranks %d\n" % (name, nr)); sys.stderr.flush()
            c = r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
            sys.stderr.write("   counting render %.2f ms, %d rays\n" % (c.render_ms, c.ray_count)); sys.stderr.flush()
        p = api.default_params(10, 1234, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS, max_spp=50)
        sys.stderr.write("== exchange %s, adaptive 10..50\n" % name); sys.stderr.flush()
        c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
        sys.stderr.write("   counting render %.2f ms, %d rays\n" % (c.render_ms, c.ray_count)); sys.stderr.flush()
