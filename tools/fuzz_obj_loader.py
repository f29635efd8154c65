"""Mutated image files / OBJ+MTL files through the host loader; run against an AddressSanitizer build of libprt_host.so
(PRT_HOST_LIB=<path>, LD_PRELOAD=libasan.so) on the CPU: nothing may crash or read out of bounds."""
import sys, os, ctypes as C, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
sys.path.insert(0, "/root/repo/tests")
from par_raytracer_amd import scenes
import texture_fixtures  # registers the textured gallery scenes
lib = C.CDLL(os.environ.get("PRT_HOST_LIB", "/root/repo/par_raytracer_amd/libprt_host.so"))
lib.prt_host_load_obj.restype = C.c_void_p
lib.prt_host_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
lib.prt_host_free_scene.argtypes = [C.c_void_p]
rng = np.random.default_rng(11)
d = tempfile.mkdtemp()
s = scenes.make_scene("textured_gallery"); scenes.write_obj(s, d, "scene.obj")
obj = open(os.path.join(d, "scene.obj"), "rb").read(); mtl = open(os.path.join(d, "scene.mtl"), "rb").read()
cp = (C.c_float * 3)(0, 0, 0)
ok = bad = 0
for it in range(3000):
    which = it % 2
    b = bytearray(obj if which == 0 else mtl)
    mode = rng.integers(0, 4)
    if mode == 0: b = b[:rng.integers(0, len(b))]
    elif mode == 1:
        for _ in range(int(rng.integers(1, 20))): b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif mode == 2:
        i = int(rng.integers(0, len(b))); j = int(rng.integers(i, min(len(b), i + 200))); del b[i:j]
    else:
        i = int(rng.integers(0, len(b))); b[i:i] = bytes(rng.integers(32, 127, size=int(rng.integers(1, 40)), dtype=np.uint8))
    open(os.path.join(d, "scene.obj" if which == 0 else "scene.mtl"), "wb").write(bytes(b))
    open(os.path.join(d, "scene.mtl" if which == 0 else "scene.obj"), "wb").write(mtl if which == 0 else obj)
    h = lib.prt_host_load_obj(d.encode(), b"scene.obj", 0, cp)
    if h: ok += 1; lib.prt_host_free_scene(h)
    else: bad += 1
print("loaded", ok, "rejected", bad)
