"""Deterministic synthetic OBJ/MTL scenes for the BASELINE.json configs.

No scene assets exist offline, so every config is generated here.  All files obey the
format constraints of the reference loader (obj_parser.cpp:130-140, 371, 387-390):
``v``/``vt``/``vn`` records, a ``g`` line before any ``f``, ``p/t/n`` index triples,
counter-clockwise front faces (the reference triangle test is single sided,
raytracer.cpp:93) and no coplanar overlapping triangles.

Floats are written with 9 significant digits so ``strtof`` recovers the exact binary32
value the generator produced.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


@dataclass
class MtlMaterial:
    """One ``newmtl`` block.  The reference callocs materials (obj_parser.cpp:255), so every
    field the shader reads must be written explicitly: a missing ``d`` means alpha 0."""
    name: str
    Ns: float = 10.0
    Ni: float = 1.5
    d: float = 1.0
    Ka: Tuple[float, float, float] = (0.75, 0.75, 0.75)
    Kd: Tuple[float, float, float] = (0.75, 0.75, 0.75)
    Ks: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    # texture maps: file names relative to the OBJ's directory (obj_parser.cpp:303-331); map_bump is a HEIGHT map
    map_Ka: Optional[str] = None
    map_Kd: Optional[str] = None
    map_Ks: Optional[str] = None
    map_d: Optional[str] = None
    map_bump: Optional[str] = None


@dataclass
class ObjGroup:
    name: str
    faces: np.ndarray            # [n,3,3] int64, 0-based (corner -> position/texcoord/normal index)
    material: Optional[str] = None


@dataclass
class ObjScene:
    name: str
    positions: np.ndarray        # [n,3] f32
    texcoords: np.ndarray        # [n,2] f32
    normals: np.ndarray          # [n,3] f32
    groups: List[ObjGroup]
    materials: List[MtlMaterial] = field(default_factory=list)
    camera_position: Tuple[float, float, float] = (0.0, 1.5, 6.0)
    camera_facing: Tuple[float, float, float] = (0.0, -0.15, -1.0)
    fov: float = 60.0
    # file name -> (uint8 image [h, w] or [h, w, c], encoding): see write_texture
    textures: Dict[str, Tuple[np.ndarray, str]] = field(default_factory=dict)
    texture_writer: Optional[object] = None      # callable(path, image, encoding) that writes one texture file

    @property
    def n_tris(self) -> int:
        return int(sum(len(g.faces) for g in self.groups))


# ----------------------------------------------------------------------------------------
# writers
# ----------------------------------------------------------------------------------------

def _fmt_rows(prefix: str, arr: np.ndarray) -> str:
    arr = np.asarray(arr, dtype=F32)
    cols = arr.shape[1]
    fmt = prefix + " " + " ".join(["%.9g"] * cols)
    return "\n".join(fmt % tuple(row) for row in arr.astype(np.float64)) + "\n"


def write_obj(scene: ObjScene, directory: str, obj_name: str = "sponza.obj") -> str:
    """Write ``scene`` as <directory>/<obj_name> (+ .mtl).  The reference binary insists on
    ``sponza.obj`` (main.cpp:553), hence the default."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, obj_name)
    mtl_name = os.path.splitext(obj_name)[0] + ".mtl"
    with open(path, "w") as f:
        f.write("# %s: %d triangles, %d groups (generated)\n" % (scene.name, scene.n_tris, len(scene.groups)))
        if scene.materials:
            f.write("mtllib %s\n" % mtl_name)
        f.write(_fmt_rows("v", scene.positions))
        f.write(_fmt_rows("vt", scene.texcoords))
        f.write(_fmt_rows("vn", scene.normals))
        for g in scene.groups:
            f.write("g %s\n" % g.name)
            if g.material is not None:
                f.write("usemtl %s\n" % g.material)
            idx = (np.asarray(g.faces, dtype=np.int64) + 1).reshape(-1, 9)
            np.savetxt(f, idx, fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
    if scene.materials:
        with open(os.path.join(directory, mtl_name), "w") as f:
            for m in scene.materials:
                f.write("newmtl %s\n" % m.name)
                f.write("Ns %.9g\nNi %.9g\nd %.9g\n" % (m.Ns, m.Ni, m.d))
                f.write("Ka %.9g %.9g %.9g\n" % tuple(m.Ka))
                f.write("Kd %.9g %.9g %.9g\n" % tuple(m.Kd))
                f.write("Ks %.9g %.9g %.9g\n" % tuple(m.Ks))
                for key in ("map_Ka", "map_Kd", "map_Ks", "map_d", "map_bump"):
                    if getattr(m, key):
                        f.write("%s %s\n" % (key, getattr(m, key)))
                f.write("\n")
    for name, (img, enc) in scene.textures.items():
        # texture files are written by whoever generated the scene (the image encoders are test fixtures: tests/texture_fixtures.py)
        if scene.texture_writer is None:
            raise ValueError("scene %s has textures but no texture_writer" % scene.name)
        scene.texture_writer(os.path.join(directory, name), img, enc)
    return path


# ----------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------

def _faces_same_index(tri_vidx: np.ndarray) -> np.ndarray:
    """[n,3] vertex ids -> [n,3,3] with position == texcoord == normal index."""
    t = np.asarray(tri_vidx, dtype=np.int64)
    return np.repeat(t[:, :, None], 3, axis=2)


def _normalize(v: np.ndarray) -> np.ndarray:
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    n[n == 0] = 1.0
    return v / n


def _grid_vertex_normals(P: np.ndarray) -> np.ndarray:
    """Smooth normals for a height-field grid P[nz+1, nx+1, 3] (up-facing)."""
    dx = np.zeros_like(P)
    dz = np.zeros_like(P)
    dx[:, 1:-1] = P[:, 2:] - P[:, :-2]
    dx[:, 0] = P[:, 1] - P[:, 0]
    dx[:, -1] = P[:, -1] - P[:, -2]
    dz[1:-1] = P[2:] - P[:-2]
    dz[0] = P[1] - P[0]
    dz[-1] = P[-1] - P[-2]
    return _normalize(np.cross(dz, dx))


# ----------------------------------------------------------------------------------------
# C1: tessellated sphere on a plane
# ----------------------------------------------------------------------------------------

def sphere_plane(segments: int = 32, rings: int = 16) -> ObjScene:
    """BASELINE config 1: UV sphere (32x16 -> 960 triangles) of radius 1 centred at (0,1,0)
    on a 20x20 plane at y=0, default material, camera (0,1.5,6) -> (0,-0.15,-1)."""
    c = np.array([0.0, 1.0, 0.0])
    pos, nrm, uv = [], [], []
    pos.append(c + [0, 1, 0]); nrm.append([0, 1, 0]); uv.append([0.5, 1.0])            # north pole
    for r in range(1, rings):
        th = np.pi * r / rings
        for s in range(segments):
            ph = 2 * np.pi * s / segments
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            pos.append(c + n); nrm.append(n); uv.append([s / segments, 1.0 - r / rings])
    pos.append(c - [0, 1, 0]); nrm.append([0, -1, 0]); uv.append([0.5, 0.0])            # south pole
    south = len(pos) - 1

    def ring(r, s):
        return 1 + (r - 1) * segments + (s % segments)

    tris = []
    for s in range(segments):
        tris.append([0, ring(1, s + 1), ring(1, s)])
    for r in range(1, rings - 1):
        for s in range(segments):
            a, b = ring(r, s), ring(r, s + 1)
            d, e = ring(r + 1, s), ring(r + 1, s + 1)
            tris.append([a, b, e])
            tris.append([a, e, d])
    for s in range(segments):
        tris.append([south, ring(rings - 1, s), ring(rings - 1, s + 1)])
    sphere_faces = _faces_same_index(np.array(tris))

    base = len(pos)
    for x, z in ((-10, -10), (10, -10), (10, 10), (-10, 10)):
        pos.append([x, 0.0, z]); nrm.append([0, 1, 0]); uv.append([(x + 10) / 20, (z + 10) / 20])
    plane_faces = _faces_same_index(np.array([[base, base + 3, base + 2], [base, base + 2, base + 1]]))

    return ObjScene(
        name="sphere_plane",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=[ObjGroup("sphere", sphere_faces), ObjGroup("plane", plane_faces)],
        camera_position=(0.0, 1.5, 6.0), camera_facing=(0.0, -0.15, -1.0), fov=60.0)


# ----------------------------------------------------------------------------------------
# C2: Cornell-box-style, 12 triangles, 6 groups, explicit MTL
# ----------------------------------------------------------------------------------------

def cornell_box() -> ObjScene:
    """BASELINE config 2: 5 wall quads + the top of a short block = 12 triangles in 6 groups.
    The left wall is low and the ceiling covers only the back half so the reference's single
    directional light (main.cpp:522-524) reaches the floor and casts the block's shadow."""
    pos, nrm, uv = [], [], []
    groups = []

    def quad(name, mat, p0, p1, p2, p3, n):
        b = len(pos)
        for p, t in zip((p0, p1, p2, p3), ((0, 0), (1, 0), (1, 1), (0, 1))):
            pos.append(p); nrm.append(n); uv.append(t)
        groups.append(ObjGroup(name, _faces_same_index(np.array([[b, b + 1, b + 2], [b, b + 2, b + 3]])), mat))

    # box interior: x in [-2,2], y in [0,4], z in [-4,0]; camera sits at z=+5 looking down -z.
    quad("floor", "white", (-2, 0, 0), (2, 0, 0), (2, 0, -4), (-2, 0, -4), (0, 1, 0))
    quad("back", "white", (-2, 0, -4), (2, 0, -4), (2, 4, -4), (-2, 4, -4), (0, 0, 1))
    quad("left", "red", (-2, 0, 0), (-2, 0, -4), (-2, 1.5, -4), (-2, 1.5, 0), (1, 0, 0))
    quad("right", "green", (2, 0, -4), (2, 0, 0), (2, 4, 0), (2, 4, -4), (-1, 0, 0))
    quad("ceiling", "white", (-2, 4, -4), (2, 4, -4), (2, 4, -2), (-2, 4, -2), (0, -1, 0))
    quad("block_top", "block", (-0.9, 1.2, -1.3), (0.5, 1.2, -1.0), (0.2, 1.2, -2.4), (-1.2, 1.2, -2.7), (0, 1, 0))

    mats = [
        MtlMaterial("white", Ns=10.0, Ni=1.5, d=1.0, Ka=(0.73, 0.73, 0.73), Kd=(0.73, 0.73, 0.73), Ks=(0.2, 0.2, 0.2)),
        MtlMaterial("red", Ns=6.0, Ni=1.45, d=1.0, Ka=(0.65, 0.05, 0.05), Kd=(0.65, 0.05, 0.05), Ks=(0.1, 0.1, 0.1)),
        MtlMaterial("green", Ns=6.0, Ni=1.45, d=1.0, Ka=(0.12, 0.45, 0.15), Kd=(0.12, 0.45, 0.15), Ks=(0.1, 0.1, 0.1)),
        MtlMaterial("block", Ns=40.0, Ni=1.8, d=1.0, Ka=(0.3, 0.35, 0.7), Kd=(0.3, 0.35, 0.7), Ks=(0.9, 0.9, 0.9)),
    ]
    return ObjScene(
        name="cornell_box",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=mats,
        camera_position=(0.0, 2.0, 5.0), camera_facing=(0.0, -0.05, -1.0), fov=60.0)


# ----------------------------------------------------------------------------------------
# C3: displaced icosphere + ground
# ----------------------------------------------------------------------------------------

def _icosphere(level: int) -> Tuple[np.ndarray, np.ndarray]:
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                  [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v = _normalize(v)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        edges_sorted = np.sort(edges, axis=1)
        uniq, inv = np.unique(edges_sorted, axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
        mid = _normalize((v[uniq[:, 0]] + v[uniq[:, 1]]) * 0.5)
        base = len(v)
        v = np.concatenate([v, mid], axis=0)
        n = len(f)
        m01, m12, m20 = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        # children stay grouped per parent face so face-cluster groups are contiguous
        f = np.stack([np.stack([a, m01, m20], 1), np.stack([b, m12, m01], 1),
                      np.stack([c, m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=1).reshape(-1, 3)
    return v, f


def displaced_icosphere(level: int = 6, n_groups: int = 64, radius: float = 2.0, amp: float = 0.18) -> ObjScene:
    """BASELINE config 3 ("Stanford-bunny-class"): icosphere level 6 = 81,920 triangles with a smooth
    radial displacement, split into ``n_groups`` contiguous face clusters, on a 2-triangle ground quad."""
    v, f = _icosphere(level)
    d = (np.sin(3.1 * v[:, 0] + 0.5) * np.cos(2.3 * v[:, 1] - 0.2) + 0.6 * np.sin(4.7 * v[:, 2] + 1.3 * v[:, 0]))
    r = radius * (1.0 + amp * d)
    p = v * r[:, None]
    # area-weighted smooth vertex normals
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    vn = np.zeros_like(p)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    vn = _normalize(vn)
    centre = np.array([0.0, radius * (1 + amp * 1.7) + 0.05, 0.0])
    p = p + centre
    uv = np.stack([0.5 + np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi), 0.5 + np.arcsin(np.clip(v[:, 1], -1, 1)) / np.pi], 1)

    groups = []
    per = len(f) // n_groups
    assert per * n_groups == len(f)
    for g in range(n_groups):
        groups.append(ObjGroup("cluster_%03d" % g, _faces_same_index(f[g * per:(g + 1) * per]), "body"))

    base = len(p)
    gp = np.array([[-12, 0, -12], [12, 0, -12], [12, 0, 12], [-12, 0, 12]], dtype=np.float64)
    p = np.concatenate([p, gp]); vn = np.concatenate([vn, np.tile([0.0, 1.0, 0.0], (4, 1))])
    uv = np.concatenate([uv, np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float64)])
    groups.append(ObjGroup("ground", _faces_same_index(np.array([[base, base + 3, base + 2], [base, base + 2, base + 1]])), "ground"))

    mats = [MtlMaterial("body", Ns=24.0, Ni=1.6, d=1.0, Ka=(0.7, 0.55, 0.4), Kd=(0.7, 0.55, 0.4), Ks=(0.6, 0.6, 0.6)),
            MtlMaterial("ground", Ns=8.0, Ni=1.4, d=1.0, Ka=(0.5, 0.55, 0.5), Kd=(0.5, 0.55, 0.5), Ks=(0.15, 0.15, 0.15))]
    return ObjScene(name="displaced_icosphere_L%d" % level,
                    positions=p.astype(F32), texcoords=uv.astype(F32), normals=vn.astype(F32),
                    groups=groups, materials=mats,
                    camera_position=(0.5, 3.6, 8.0), camera_facing=(-0.05, -0.12, -1.0), fov=60.0)


def many_materials(level: int = 3, n_groups: int = 40) -> ObjScene:
    """The displaced icosphere with one material per face cluster (41 MTL materials + the scene default = 42): more
    than the 32 the shading kernels stage in LDS, so the global-table path runs.  Every fifth cluster is translucent."""
    s = displaced_icosphere(level, n_groups)
    rng = np.random.default_rng(77)
    mats = [m for m in s.materials if m.name == "ground"]
    for g in range(n_groups):
        kd = tuple(float(v) for v in rng.uniform(0.25, 0.95, size=3))
        mats.append(MtlMaterial("m%02d" % g, Ns=float(rng.uniform(4.0, 80.0)), Ni=float(rng.uniform(1.1, 2.2)),
                                d=0.6 if g % 5 == 2 else 1.0, Ka=kd, Kd=kd, Ks=tuple(float(v) for v in rng.uniform(0.0, 0.9, size=3))))
        s.groups[g].material = "m%02d" % g
    s.materials = mats
    s.name = "many_materials_L%d" % level
    return s


# ----------------------------------------------------------------------------------------
# C4 / C5: height-field terrain with canyon walls
# ----------------------------------------------------------------------------------------

def terrain(quads: int = 708, tiles: int = 32, size: float = 708.0, seed: int = 7) -> ObjScene:
    """BASELINE configs 4-5: ``quads x quads`` height-field (708 -> 1,002,528 triangles) split into
    ``tiles x tiles`` groups (32 -> 1,024).  A winding canyon with steep walls keeps bounce rays inside
    the scene instead of escaping to the sky."""
    n = quads
    rng = np.random.RandomState(seed)
    xs = np.linspace(-size / 2, size / 2, n + 1)
    zs = np.linspace(-size / 2, size / 2, n + 1)
    X, Z = np.meshgrid(xs, zs)                       # [z, x]
    u, w = X / size, Z / size
    h = np.zeros_like(X)
    for octave in range(6):                         # deterministic value-noise-like sum of sines
        fx, fz = rng.uniform(1.5, 4.0, 2) * (1.9 ** octave)
        px, pz = rng.uniform(0, 2 * np.pi, 2)
        rot = rng.uniform(0, np.pi)
        a = np.cos(rot) * u + np.sin(rot) * w
        b = -np.sin(rot) * u + np.cos(rot) * w
        h += (0.5 ** octave) * np.sin(2 * np.pi * fx * a + px) * np.cos(2 * np.pi * fz * b + pz)
    h *= size * 0.035
    # canyon: a sinuous channel along z, ~6% of the width, with near-vertical walls
    centre = 0.12 * np.sin(2 * np.pi * 1.5 * w + 0.7) + 0.05 * np.sin(2 * np.pi * 4.0 * w)
    dist = np.abs(u - centre)
    half = 0.035
    wall = 1.0 / (1.0 + np.exp(-(dist - half) / 0.004))   # 0 inside, 1 outside
    h = h * (0.35 + 0.65 * wall) + size * 0.11 * wall
    P = np.stack([X, h, Z], axis=-1)
    N = _grid_vertex_normals(P)
    uv = np.stack([(X / size + 0.5) * 16.0, (Z / size + 0.5) * 16.0], axis=-1)

    vid = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    groups = []
    edges = np.linspace(0, n, tiles + 1).astype(int)
    for tz in range(tiles):
        for tx in range(tiles):
            z0, z1, x0, x1 = edges[tz], edges[tz + 1], edges[tx], edges[tx + 1]
            a = vid[z0:z1, x0:x1].reshape(-1)
            b = vid[z0:z1, x0 + 1:x1 + 1].reshape(-1)
            c = vid[z0 + 1:z1 + 1, x0 + 1:x1 + 1].reshape(-1)
            d = vid[z0 + 1:z1 + 1, x0:x1].reshape(-1)
            # up-facing CCW: (a, d, c) and (a, c, b) with x to the right and z toward the viewer
            tri = np.stack([np.stack([a, d, c], 1), np.stack([a, c, b], 1)], axis=1).reshape(-1, 3)
            mat = "rock" if (tx + tz) % 2 == 0 else "soil"
            groups.append(ObjGroup("tile_%02d_%02d" % (tz, tx), _faces_same_index(tri), mat))

    mats = [MtlMaterial("rock", Ns=18.0, Ni=1.55, d=1.0, Ka=(0.55, 0.5, 0.45), Kd=(0.55, 0.5, 0.45), Ks=(0.35, 0.35, 0.35)),
            MtlMaterial("soil", Ns=6.0, Ni=1.35, d=1.0, Ka=(0.45, 0.5, 0.3), Kd=(0.45, 0.5, 0.3), Ks=(0.1, 0.1, 0.1))]
    # camera inside the canyon mouth, looking up the channel and slightly down
    zc = size * 0.46
    cx = float((0.12 * np.sin(2 * np.pi * 1.5 * 0.46 + 0.7) + 0.05 * np.sin(2 * np.pi * 4.0 * 0.46)) * size)
    return ObjScene(name="terrain_%dx%d" % (n, n),
                    positions=P.reshape(-1, 3).astype(F32), texcoords=uv.reshape(-1, 2).astype(F32),
                    normals=N.reshape(-1, 3).astype(F32), groups=groups, materials=mats,
                    camera_position=(cx, size * 0.075, zc), camera_facing=(-0.12, -0.22, -1.0), fov=60.0)


def _quad(pos, nrm, uv, corners, normal, uvs):
    """Append a quad (4 corners, counter-clockwise seen from `normal`) -> two triangles of vertex ids."""
    base = len(pos)
    for c, t in zip(corners, uvs):
        pos.append(list(c)); nrm.append(list(normal)); uv.append(list(t))
    return [[base, base + 1, base + 2], [base, base + 2, base + 3]]


def coincident_geometry() -> ObjScene:
    """Adversarial geometry for the one place where the reference's VISIT ORDER is observable (raytracer.cpp:104, 149,
    208-209, 220): hits whose t agree to within a few ulp.  Real OBJ scenes have them - decals laid onto walls, faces
    exported twice, coplanar patches from different groups - and the reference resolves them by a sequential filter
    (`t > best * d` early reject, then strict `<`) in sphere-tree order, which any other traversal order has to replay.

    A slanted base plane (so that t, d and the barycentrics round differently for every triangulation) carries, each in
    its own group and material: a coplanar patch triangulated the other way; patches lifted by 1, 2, 3 and 4 ulp of the
    coordinates; a patch pushed 2 ulp BELOW the base; a patch whose faces appear twice in the group and once more in
    another group; a finely tessellated coplanar patch (32 x 32 quads: shared edges and vertices hit head-on); a
    translucent coplanar patch (a different winner changes the ray count, not only the colour).  A second slanted wall
    catches the bounces."""
    pos, nrm, uv = [], [], []
    groups = []
    n = _normalize(np.array([[0.2, 1.0, 0.35]], dtype=np.float64))[0]
    u = _normalize(np.array([[1.0, 0.3, 0.1]], dtype=np.float64))[0]
    u = _normalize((u - n * np.dot(u, n))[None, :])[0]
    v = np.cross(n, u)
    origin = np.array([0.3, 0.9, -1.7])

    def P(a, b, lift_ulps=0.0):
        q = origin + a * u + b * v
        q32 = q.astype(np.float32)
        if lift_ulps:
            # move every coordinate by k ulp (of itself) along the sign of the plane normal's component
            q32 = (q32.astype(np.float64) + lift_ulps * np.spacing(np.abs(q32)).astype(np.float64) * np.sign(n)).astype(np.float32)
        return [float(x) for x in q32]

    def patch(name, material, a0, a1, b0, b1, lift=0.0, flip_diag=False, repeat=1, cells=1):
        tris = []
        for r in range(repeat):
            for ia in range(cells):
                for ib in range(cells):
                    aa0 = a0 + (a1 - a0) * ia / cells; aa1 = a0 + (a1 - a0) * (ia + 1) / cells
                    bb0 = b0 + (b1 - b0) * ib / cells; bb1 = b0 + (b1 - b0) * (ib + 1) / cells
                    base = len(pos)
                    for (a, b) in ((aa0, bb0), (aa1, bb0), (aa1, bb1), (aa0, bb1)):     # counter-clockwise seen from +n
                        pos.append(P(a, b, lift)); nrm.append([float(x) for x in n]); uv.append([a, b])
                    if flip_diag:
                        tris += [[base + 1, base + 2, base + 3], [base + 1, base + 3, base]]
                    else:
                        tris += [[base, base + 1, base + 2], [base, base + 2, base + 3]]
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))
        return tris

    patch("base", "base", -4.0, 4.0, -4.0, 4.0)
    patch("coplanar_flip", "m_flip", -3.5, -2.0, -3.0, -1.0, flip_diag=True)
    for k in (1, 2, 3, 4):
        patch("lift_%dulp" % k, "m_lift%d" % k, -1.8 + 0.9 * (k - 1), -1.0 + 0.9 * (k - 1), -3.0, -1.0, lift=float(k))
    patch("sunk_2ulp", "m_sunk", 2.0, 3.5, -3.0, -1.0, lift=-2.0)
    t_double = patch("doubled", "m_double", -3.5, -1.5, -0.5, 1.0, repeat=2)
    # the same faces once more, through another group with another material (re-using the first copy's vertices)
    groups.append(ObjGroup("doubled_again", _faces_same_index(np.array(t_double[:2])), "m_double2"))
    patch("tessellated", "m_tess", -1.0, 1.5, -0.5, 2.0, cells=32)
    patch("translucent", "m_trans", 2.0, 3.5, -0.5, 1.5)
    patch("stack_a", "m_lift1", -3.5, -1.5, 1.5, 3.5, lift=1.0)
    patch("stack_b", "m_lift2", -3.0, -1.0, 2.0, 3.8, lift=1.0, flip_diag=True)
    patch("stack_c", "m_lift3", -2.5, -0.5, 1.2, 3.0)
    # a wall facing the base plane and the light, for the bounces
    wn = _normalize(np.array([[-0.6, 0.25, 0.75]]))[0]
    wu = _normalize(np.cross(wn, [0.0, 1.0, 0.0])[None, :])[0]
    wv = np.cross(wu, wn)
    wo = np.array([3.2, 1.2, -3.2])
    corners = [wo - 2.5 * wu - 1.2 * wv, wo + 2.5 * wu - 1.2 * wv, wo + 2.5 * wu + 2.4 * wv, wo - 2.5 * wu + 2.4 * wv]
    if np.dot(np.cross(corners[1] - corners[0], corners[2] - corners[0]), wn) < 0:
        corners = corners[::-1]
    groups.append(ObjGroup("wall", _faces_same_index(np.array(_quad(pos, nrm, uv, corners, tuple(wn), [(0, 0), (1, 0), (1, 1), (0, 1)]))), "wall"))

    def mat(name, kd, d=1.0, ks=(0.2, 0.2, 0.2), ns=12.0):
        return MtlMaterial(name, Ns=ns, Ni=1.5, d=d, Ka=kd, Kd=kd, Ks=ks)

    materials = [
        mat("base", (0.55, 0.55, 0.55)), mat("m_flip", (0.9, 0.1, 0.1)), mat("m_lift1", (0.1, 0.8, 0.1)), mat("m_lift2", (0.1, 0.1, 0.9)),
        mat("m_lift3", (0.9, 0.8, 0.1)), mat("m_lift4", (0.8, 0.1, 0.8)), mat("m_sunk", (0.1, 0.8, 0.8)),
        mat("m_double", (1.0, 0.5, 0.0), ks=(0.8, 0.8, 0.8), ns=40.0), mat("m_double2", (0.0, 0.4, 1.0)),
        mat("m_tess", (0.9, 0.9, 0.9), ks=(0.5, 0.5, 0.5), ns=25.0), mat("m_trans", (0.9, 0.3, 0.5), d=0.4),
        mat("wall", (0.7, 0.65, 0.5)),
    ]
    return ObjScene(
        name="coincident_geometry",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials,
        camera_position=(0.2, 5.2, 4.4), camera_facing=(0.02, -0.78, -0.9), fov=60.0)


# ----------------------------------------------------------------------------------------
# registry: name -> (factory, render defaults)
# ----------------------------------------------------------------------------------------

@dataclass(frozen=True)
class RenderConfig:
    scene: str
    width: int
    height: int
    spp: int
    bounce_depth: int = 2
    seed: int = 1234


SCENES = {
    "sphere_plane": lambda: sphere_plane(),
    "cornell_box": lambda: cornell_box(),
    "icosphere_l6": lambda: displaced_icosphere(6, 64),
    "icosphere_l3": lambda: displaced_icosphere(3, 16),       # 1,280 tris, CPU-test sized
    "terrain_1m": lambda: terrain(708, 32),
    "terrain_64": lambda: terrain(64, 4, size=64.0),          # 8,192 tris in 16 groups, CPU-test sized
    "terrain_192": lambda: terrain(192, 8, size=192.0),       # 73,728 tris in 64 groups
    "many_materials": lambda: many_materials(),               # 1,282 tris, 42 materials (LDS table fallback), translucent clusters
    "coincident": lambda: coincident_geometry(),              # 2,100 tris: coplanar / ulp-offset / doubled faces in different groups (visit-order parity)
}

CONFIGS: Dict[str, RenderConfig] = {
    "C1": RenderConfig("sphere_plane", 256, 256, 1),
    "C2": RenderConfig("cornell_box", 512, 512, 4),
    "C3": RenderConfig("icosphere_l6", 1920, 1080, 8),
    "C4": RenderConfig("terrain_1m", 1920, 1080, 8),
    "C5": RenderConfig("terrain_1m", 3840, 2160, 64, bounce_depth=8),
}


def register_scene(name: str, factory) -> None:
    """Adds a generator to the registry (the textured fixture scenes of tests/texture_fixtures.py register themselves)."""
    SCENES[name] = factory


def make_scene(name: str) -> ObjScene:
    return SCENES[name]()
