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
        write_texture(os.path.join(directory, name), img, enc)
    return path


# ----------------------------------------------------------------------------------------
# texture image writers (test inputs for the loader's decoders; every variant the decoders accept)
# ----------------------------------------------------------------------------------------

def _png_chunk(tag: bytes, body: bytes) -> bytes:
    import struct, zlib
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def _png_filter_rows(raw: np.ndarray, bpp: int, cycle=(0, 1, 2, 3, 4)) -> bytes:
    """raw: [h, stride] uint8 scanlines.  Rows cycle through the five PNG filter types so a decoder has to
    implement all of them (None, Sub, Up, Average, Paeth), or through `cycle`."""
    h, stride = raw.shape
    out = bytearray()
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        cur = raw[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, dtype=np.int32), cur[:-bpp]]) if stride > bpp else np.zeros(stride, dtype=np.int32)
        b = prev
        c = np.concatenate([np.zeros(bpp, dtype=np.int32), prev[:-bpp]]) if stride > bpp else np.zeros(stride, dtype=np.int32)
        ft = cycle[y % len(cycle)]
        if ft == 0:
            pred = np.zeros(stride, dtype=np.int32)
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) >> 1
        else:
            pp = a + b - c
            pa, pb, pc = np.abs(pp - a), np.abs(pp - b), np.abs(pp - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        out.append(ft)
        out += ((cur - pred) & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _png_pack_bits(rows: np.ndarray, bits: int) -> np.ndarray:
    """[h, n] sample values below 2^bits -> [h, ceil(n * bits / 8)] bytes, first sample in the high bits."""
    h, n = rows.shape
    per = 8 // bits
    padded = np.zeros((h, (n + per - 1) // per * per), dtype=np.uint8)
    padded[:, :n] = rows
    out = np.zeros((h, padded.shape[1] // per), dtype=np.uint8)
    for k in range(per):
        out |= (padded[:, k::per] << (8 - bits * (k + 1))).astype(np.uint8)
    return out


def write_png(path: str, img: np.ndarray, sixteen_bit: bool = False, palette: bool = False, palette_alpha: bool = False,
              interlace: bool = False, bits: int = 8, key=None, filters=None) -> None:
    """8-bit grey / grey+alpha / RGB / RGBA PNG; `sixteen_bit` stores every sample as (v, 255 - v) big endian
    (a decoder keeping the high byte recovers v); `palette` quantises an RGB(A) image to <= 256 colours.
    `bits` 1 / 2 / 4: a grey image whose values are already below 2^bits, or palette indices packed that tightly.
    `interlace`: Adam7, each of the seven passes filtered on its own.  `key`: a tRNS chunk naming one transparent
    grey level / RGB colour (in the file's own sample values)."""
    import struct, zlib
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    chunks = b""
    if palette:
        flat = a.reshape(-1, c)
        colours, index = np.unique(flat, axis=0, return_inverse=True)
        assert len(colours) <= (1 << bits), "too many colours for the palette"
        ctype, depth = 3, bits
        samples = index.reshape(h, w, 1).astype(np.uint8)
        chunks += _png_chunk(b"PLTE", colours[:, :3].astype(np.uint8).tobytes())
        if palette_alpha:
            assert c == 4
            chunks += _png_chunk(b"tRNS", colours[:, 3].astype(np.uint8).tobytes())
    else:
        ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
        if sixteen_bit:
            depth = 16
            samples = np.stack([a, 255 - a], axis=-1).reshape(h, w, c * 2)
        else:
            depth = bits
            assert bits == 8 or c == 1
            samples = a
        if key is not None:
            kv = [int(v) for v in np.atleast_1d(key)]
            assert len(kv) == c and c in (1, 3)
            chunks += _png_chunk(b"tRNS", b"".join(struct.pack(">H", (v << 8 | (255 - v)) if sixteen_bit else v) for v in kv))
    bpp = max(1, samples.shape[2] * (depth if depth < 8 else 8) // 8)     # the filters' "bytes per pixel"

    def filtered(sub):                                                  # sub: [ph, pw, bytes or samples per pixel]
        ph, pw = sub.shape[:2]
        rows = sub.reshape(ph, -1)
        if depth < 8:
            rows = _png_pack_bits(rows, depth)
        # Sub-byte samples: filters None and Sub only, unless asked otherwise.  The reference's decoder (stb_image 2.14) reads
        # the previous row of such images at the wrong offset - partly memory it never wrote - so files that use Up /
        # Average / Paeth there have no reference answer to compare with (real encoders default to None for them).
        return _png_filter_rows(np.ascontiguousarray(rows), bpp, filters if filters else ((0, 1) if depth < 8 else (0, 1, 2, 3, 4)))

    if interlace:
        body = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                body += filtered(sub)
    else:
        body = filtered(samples)
    data = zlib.compress(body, 6)
    # two IDAT chunks: a decoder must concatenate them
    cut = len(data) // 2
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0)))
        f.write(chunks)
        f.write(_png_chunk(b"IDAT", data[:cut]))
        f.write(_png_chunk(b"IDAT", data[cut:]))
        f.write(_png_chunk(b"IEND", b""))


def write_tga(path: str, img: np.ndarray, rle: bool = False, top_down: bool = False, kind: str = "") -> None:
    """Grey (type 3 / 11), BGR or BGRA (type 2 / 10) TGA; bottom-up unless `top_down` (descriptor bit 5).
    kind: "" (8 / 24 / 32 bits from the channel count), "16" (5-5-5 pixels), "ga" (2 channels: 16-bit grey + alpha, type
    3), "map24" / "map32" / "map16" (colour map with 24- / 32- / 15-bit entries, 8-bit indices; with an image id and a
    non-zero first-entry field), "map24_i16" (16-bit indices)."""
    import struct
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    rows = a if top_down else a[::-1]
    ident = b""
    cmap = b""
    cmap_spec = (0, 0, 0)

    def bgr(x):
        return x[..., [2, 1, 0] + ([3] if x.shape[-1] == 4 else [])] if x.shape[-1] >= 3 else x

    def pack555(x):
        v = ((x[..., 0].astype(np.uint16) >> 3) << 10) | ((x[..., 1].astype(np.uint16) >> 3) << 5) | (x[..., 2].astype(np.uint16) >> 3)
        return v.astype("<u2")

    if kind == "":
        assert c in (1, 3, 4)
        px = np.ascontiguousarray(bgr(rows)).reshape(-1, c)
        itype, bits, abits = (3 if c == 1 else 2), 8 * c, (8 if c == 4 else 0)
    elif kind == "16":
        px = pack555(rows).reshape(-1, 1).view(np.uint8).reshape(-1, 2)
        itype, bits, abits = 2, 16, 0
    elif kind == "ga":
        assert c == 2
        px = np.ascontiguousarray(rows).reshape(-1, 2)
        itype, bits, abits = 3, 16, 8
    elif kind.startswith("map"):
        flat = rows.reshape(-1, c)
        colours, index = np.unique(flat, axis=0, return_inverse=True)
        ident = b"id!"                                                  # an image id to skip
        first = 5                                                       # "first entry index": the reference's decoder skips that many BYTES
        if kind == "map16":
            table = pack555(colours[:, :3]).view(np.uint8).reshape(-1, 2)
            ebits = 15
        else:
            table = bgr(colours)
            ebits = 8 * colours.shape[1]
        cmap = bytes(first) + np.ascontiguousarray(table).tobytes()
        cmap_spec = (first, len(colours), ebits)
        if kind.endswith("_i16"):
            px = index.astype("<u2").reshape(-1, 1).view(np.uint8).reshape(-1, 2)
            bits = 16
        else:
            assert len(colours) <= 256
            px = index.astype(np.uint8).reshape(-1, 1)
            bits = 8
        itype, abits = 1, 0
    else:
        raise ValueError(kind)
    c = px.shape[1]
    header = struct.pack("<BBBHHBHHHHBB", len(ident), 1 if cmap else 0, itype + (8 if rle else 0), cmap_spec[0], cmap_spec[1], cmap_spec[2],
                         0, 0, w, h, bits, (0x20 if top_down else 0) | abits)
    body = bytearray()
    if not rle:
        body += px.tobytes()
    else:
        i, n = 0, len(px)
        while i < n:
            run = 1
            while i + run < n and run < 128 and np.array_equal(px[i + run], px[i]):
                run += 1
            if run >= 2:
                body.append(0x80 | (run - 1))
                body += px[i].tobytes()
                i += run
            else:
                j = i + 1
                while j < n and j - i < 128 and not (j + 1 < n and np.array_equal(px[j], px[j + 1])):
                    j += 1
                body.append(j - i - 1)
                body += px[i:j].tobytes()
                i = j
    with open(path, "wb") as f:
        f.write(header)
        f.write(ident)
        f.write(cmap)
        f.write(bytes(body))


def write_bmp(path: str, img: np.ndarray, kind: str = "24") -> None:
    """Uncompressed BMP.  kind: "24" (bottom-up BGR), "24_top" (negative height), "os2_24" / "os2_8" (12-byte header; the
    palette has 3-byte entries), "8" / "4" (palette, <= 256 / 16 colours), "16" (5-5-5), "16_565" (40-byte header +
    BITFIELDS masks), "32" (plain BGRA: img may have 4 channels), "32_v4" (108-byte header with R, G, B, A masks in an
    unusual order)."""
    import struct
    a = np.asarray(img, dtype=np.uint8)
    h, w, c = a.shape
    top = kind == "24_top"
    src = a if top else a[::-1]
    masks = b""
    pal = b""
    comp = 0
    hsz = 12 if kind.startswith("os2") else (108 if kind == "32_v4" else 40)
    if kind in ("24", "24_top", "os2_24"):
        bpp = 24
        body = src[:, :, 2::-1].reshape(h, w * 3)
    elif kind in ("8", "4", "os2_8"):
        bpp = 4 if kind == "4" else 8
        colours, index = np.unique(a[:, :, :3].reshape(-1, 3), axis=0, return_inverse=True)
        assert len(colours) <= (1 << bpp)
        idx = index.reshape(h, w)[::-1].astype(np.uint8)
        body = _png_pack_bits(idx, 4) if bpp == 4 else idx
        for col in colours:
            pal += bytes([int(col[2]), int(col[1]), int(col[0])]) + (b"" if hsz == 12 else b"\0")
        pal += bytes((3 if hsz == 12 else 4) * ((1 << bpp) - len(colours)))        # a full-size palette, as real files carry
    elif kind in ("16", "16_565"):
        bpp = 16
        r, g, b_ = (src[:, :, k].astype(np.uint16) for k in range(3))
        if kind == "16":
            v = ((r >> 3) << 10) | ((g >> 3) << 5) | (b_ >> 3)
        else:
            v = ((r >> 3) << 11) | ((g >> 2) << 5) | (b_ >> 3)
            comp = 3
            masks = struct.pack("<III", 0xF800, 0x07E0, 0x001F)
        body = v.astype("<u2").view(np.uint8).reshape(h, w * 2)
    elif kind == "32":
        bpp = 32
        alpha = src[:, :, 3] if c == 4 else np.zeros((h, w), dtype=np.uint8)
        body = np.stack([src[:, :, 2], src[:, :, 1], src[:, :, 0], alpha], axis=2).reshape(h, w * 4)
    elif kind == "32_v4":
        bpp = 32
        comp = 3
        alpha = src[:, :, 3] if c == 4 else np.full((h, w), 255, dtype=np.uint8)
        body = np.stack([alpha, src[:, :, 0], src[:, :, 1], src[:, :, 2]], axis=2).reshape(h, w * 4)       # bytes A R G B = masks below
    else:
        raise ValueError(kind)
    stride = (body.shape[1] + 3) & ~3
    rows = np.zeros((h, stride), dtype=np.uint8)
    rows[:, :body.shape[1]] = body
    offset = 14 + hsz + len(masks) + len(pal)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", offset + stride * h, 0, 0, offset))
        if hsz == 12:
            f.write(struct.pack("<IHHHH", 12, w, h, 1, bpp))
        else:
            f.write(struct.pack("<IiiHHIIiiII", hsz, w, -h if top else h, 1, bpp, comp, stride * h, 2835, 2835, 0, 0))
            if hsz == 108:
                f.write(struct.pack("<IIII", 0x0000FF00, 0x00FF0000, 0xFF000000, 0x000000FF))     # R, G, B, A masks
                f.write(b"BGRs" + b"\0" * 48)                                                       # colour space + endpoints + gammas
        f.write(masks)
        f.write(pal)
        f.write(rows.tobytes())


def write_gif(path: str, img: np.ndarray, interlace: bool = False, transparent: bool = False, local_table: bool = False,
              canvas: Optional[Tuple[int, int, int, int]] = None, bgindex: int = 0) -> None:
    """GIF89a with one image.  img: [h, w, 3] with at most 256 colours (255 with `transparent`: pixels equal to img[0, 0]
    then get the transparent index of a graphic control extension).  `canvas` = (W, H, x, y): the image sits at (x, y) on a
    larger logical screen.  `local_table`: the palette is the image's local colour table (a 2-entry global one remains)."""
    import struct
    a = np.asarray(img, dtype=np.uint8)
    h, w, _ = a.shape
    colours, index = np.unique(a.reshape(-1, 3), axis=0, return_inverse=True)
    index = index.reshape(h, w)
    n = len(colours)
    assert n <= 256
    bits = max(1, (max(n, 2) - 1).bit_length())
    table = np.zeros((1 << bits, 3), dtype=np.uint8)
    table[:n] = colours
    W, H, x0, y0 = canvas if canvas else (w, h, 0, 0)
    out = bytearray(b"GIF89a")
    if local_table:
        out += struct.pack("<HHBBB", W, H, 0x80 | 0, bgindex & 1, 0) + bytes([200, 30, 90, 10, 220, 140])       # 2-entry global table
    else:
        out += struct.pack("<HHBBB", W, H, 0x80 | (bits - 1), bgindex, 0) + table.tobytes()
    out += b"\x21\xFE\x05hello\x00"                                                                                  # a comment extension
    if transparent:
        out += b"\x21\xF9\x04" + bytes([0x01, 0, 0, int(index[0, 0])]) + b"\x00"
    out += b"\x2C" + struct.pack("<HHHHB", x0, y0, w, h, (0x40 if interlace else 0) | ((0x80 | (bits - 1)) if local_table else 0))
    if local_table:
        out += table.tobytes()
    rows = list(range(h))
    if interlace:
        rows = list(range(0, h, 8)) + list(range(4, h, 8)) + list(range(2, h, 4)) + list(range(1, h, 2))
    data = index[rows].reshape(-1)
    # LZW, variable code size, codes packed least significant bit first
    min_cs = max(2, bits)
    clear, eoi = 1 << min_cs, (1 << min_cs) + 1
    acc, nacc, stream = 0, 0, bytearray()

    def put(code, size):
        nonlocal acc, nacc
        acc |= code << nacc
        nacc += size
        while nacc >= 8:
            stream.append(acc & 0xFF)
            acc >>= 8
            nacc -= 8

    size = min_cs + 1
    dictionary = {}
    nxt = eoi + 1
    put(clear, size)
    prefix = int(data[0])
    for v in data[1:]:
        v = int(v)
        key = (prefix, v)
        if key in dictionary:
            prefix = dictionary[key]
            continue
        put(prefix, size)
        if nxt < 4096:
            dictionary[key] = nxt
            nxt += 1
            if nxt - 1 == (1 << size) and size < 12:
                size += 1
        else:
            put(clear, size)
            dictionary = {}
            nxt = eoi + 1
            size = min_cs + 1
        prefix = v
    put(prefix, size)
    put(eoi, size)
    if nacc:
        stream.append(acc & 0xFF)
    out.append(min_cs)
    for i in range(0, len(stream), 255):
        chunk = stream[i:i + 255]
        out.append(len(chunk))
        out += chunk
    out += b"\x00\x3B"
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_psd(path: str, img: np.ndarray, rle: bool = False, sixteen_bit: bool = False) -> None:
    """Photoshop file with only the flattened composite: RGB mode, img [h, w, 3 or 4], planar channels, raw or PackBits
    rows; `sixteen_bit` stores every sample as (v, 255 - v)."""
    import struct
    a = np.asarray(img, dtype=np.uint8)
    h, w, c = a.shape
    out = bytearray(b"8BPS" + struct.pack(">H6xHIIHH", 1, c, h, w, 16 if sixteen_bit else 8, 3))
    out += struct.pack(">I", 0)                                         # colour mode data
    out += struct.pack(">I", 12) + b"8BIM\x03\xed\x00\x00\x00\x00\x00\x00"   # one (empty) image resource
    out += struct.pack(">I", 0)                                         # layers and masks
    out += struct.pack(">H", 1 if rle else 0)
    if not rle:
        for k in range(c):
            plane = a[:, :, k]
            out += (np.stack([plane, 255 - plane], axis=-1).tobytes() if sixteen_bit else plane.tobytes())
    else:
        assert not sixteen_bit
        rows = []
        for k in range(c):
            for y in range(h):
                row, packed, i = a[y, :, k], bytearray(), 0
                while i < w:
                    run = 1
                    while i + run < w and run < 128 and row[i + run] == row[i]:
                        run += 1
                    if run >= 3:
                        packed += bytes([257 - run, int(row[i])])
                        i += run
                    else:
                        j = i
                        while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                            j += 1
                        packed += bytes([j - i - 1]) + row[i:j].tobytes()
                        i = j
                rows.append(bytes(packed))
        out += b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows)
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_hdr(path: str, img: np.ndarray, rle: bool = True, magic: str = "#?RADIANCE") -> None:
    """Radiance RGBE picture.  img: [h, w, 3] floats (radiance) or uint8 (mapped to a few decades of radiance).  `rle`:
    new-style run-length scanlines (only legal for 8 <= w < 32768), else flat RGBE pixels."""
    a = np.asarray(img)
    if a.dtype == np.uint8:
        a = (a.astype(np.float64) / 255.0) ** 2.2 * 4.0 + (a.astype(np.float64) % 7 == 0) * 1e-4
    a = a.astype(np.float64)
    h, w, _ = a.shape
    m = a.max(axis=2)
    mant, expo = np.frexp(np.maximum(m, 1e-38))
    scale = np.where(m < 1e-32, 0.0, mant * 256.0 / np.maximum(m, 1e-38))
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    rgbe[:, :, :3] = np.clip(a * scale[:, :, None], 0, 255).astype(np.uint8)
    rgbe[:, :, 3] = np.where(m < 1e-32, 0, expo + 128).astype(np.uint8)
    rgbe[m < 1e-32] = 0
    out = bytearray((magic + "\n# made by par_raytracer_amd.scenes\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w)).encode())
    if not rle:
        out += rgbe.tobytes()
    else:
        assert 8 <= w < 32768
        for y in range(h):
            out += bytes([2, 2, w >> 8, w & 255])
            for k in range(4):
                row, i = rgbe[y, :, k], 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and row[i + run] == row[i]:
                        run += 1
                    if run >= 3:
                        out += bytes([128 + run, int(row[i])])
                        i += run
                    else:
                        j = i
                        while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                            j += 1
                        out += bytes([j - i]) + row[i:j].tobytes()
                        i = j
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_pic(path: str, img: np.ndarray, kind: str = "mixed") -> None:
    """Softimage PIC, 8 bits per channel.  img [h, w, 3 or 4]; one packet for R, G, B (and, with 4 channels, a second one
    for alpha).  kind: "raw", "pure" (count, value pairs) or "mixed" (runs and literal stretches; one long run uses the
    16-bit count form)."""
    import struct
    a = np.asarray(img, dtype=np.uint8)
    h, w, c = a.shape
    ptype = {"raw": 0, "pure": 1, "mixed": 2}[kind]
    out = bytearray(b"\x53\x80\xF6\x34" + struct.pack(">f", 3.71) + b"par_raytracer_amd".ljust(80, b"\0") + b"PICT")
    out += struct.pack(">HHfHH", w, h, 1.0, 3, 0)
    out += bytes([1 if c == 4 else 0, 8, ptype, 0xE0])
    if c == 4:
        out += bytes([0, 8, ptype, 0x10])

    def encode(row):                                                    # row: [w, n] values of one packet
        n = row.shape[1]
        if ptype == 0:
            return row.tobytes()
        body, i = bytearray(), 0
        while i < w:
            run = 1
            while i + run < w and run < (255 if ptype == 1 else 60000) and np.array_equal(row[i + run], row[i]):
                run += 1
            if ptype == 1:
                body += bytes([run]) + row[i].tobytes()
                i += run
            elif run >= 2:
                body += (bytes([run + 127]) if run <= 128 else bytes([128]) + struct.pack(">H", run)) + row[i].tobytes()
                i += run
            else:
                j = i + 1
                while j < w and j - i < 128 and not (j + 1 < w and np.array_equal(row[j], row[j + 1])):
                    j += 1
                body += bytes([j - i - 1]) + row[i:j].tobytes()
                i = j
        return bytes(body)

    for y in range(h):
        out += encode(a[y, :, :3])
        if c == 4:
            out += encode(a[y, :, 3:4])
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_pnm(path: str, img: np.ndarray) -> None:
    """Binary PGM (grey) / PPM (RGB) with a comment line in the header."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    assert c in (1, 3)
    with open(path, "wb") as f:
        f.write(("P%d\n# generated\n%d %d\n255\n" % (5 if c == 1 else 6, w, h)).encode())
        f.write(np.ascontiguousarray(a).tobytes())


_JPEG_ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _jpeg_huffman_table(freq: dict):
    """Code lengths (1..16) for the symbols of `freq` by the procedure of ITU T.81 Annex K.2: Huffman's algorithm with one
    reserved symbol (so that no code is all ones), lengths above 16 folded back.  Returns (counts[16], symbols in code order,
    {symbol: (code, length)})."""
    import heapq
    items = [(f, 0, [sym]) for sym, f in freq.items() if f > 0]
    items.append((0, 1, [256]))                                        # the reserved symbol: least frequent, longest code
    depth = {sym: 0 for _, _, syms in items for sym in syms}
    heap = [(f, tie, i, syms) for i, (f, tie, syms) in enumerate(items)]
    heapq.heapify(heap)
    n = len(heap)
    if n == 1:
        depth[heap[0][3][0]] = 1
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        for sym in a[3] + b[3]:
            depth[sym] += 1
        n += 1
        heapq.heappush(heap, (a[0] + b[0], max(a[1], b[1]), n, a[3] + b[3]))
    bits = [0] * 64
    for sym, d in depth.items():
        bits[d] += 1
    for i in range(63, 16, -1):                                         # K.2 figure K.3: fold lengths above 16 back
        while bits[i] > 0:
            j = i - 2
            while bits[j] == 0:
                j -= 1
            bits[i] -= 2
            bits[i - 1] += 1
            bits[j + 1] += 2
            bits[j] -= 1
    i = 16
    while bits[i] == 0:
        i -= 1
    bits[i] -= 1                                                        # drop the reserved symbol's code
    order = sorted((sym for sym in depth if sym != 256), key=lambda sym: (depth[sym], sym))
    counts = bits[1:17]
    codes, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            codes[order[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    assert k == len(order)
    return counts, order, codes


class _JpegBits:
    """Entropy-coded segment writer: MSB-first bits, 0xFF byte stuffing, 1-padding before markers."""
    def __init__(self):
        self.data = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.data.append(byte)
            if byte == 0xFF:
                self.data.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def pad(self):
        if self.n % 8:
            self.put((1 << (8 - self.n % 8)) - 1, 8 - self.n % 8)

    def marker(self, m):
        self.pad()
        self.data += bytes([0xFF, m])


def _write_progressive_jpeg(path, H, W, comps, qtabs, mcus_x, mcus_y, restart, rgb_ids, scan_blocks) -> None:
    """ITU T.81 Annex G.  Scan script: DC of all components (point transform 1); per component AC 1..5 and 6..63 (point
    transform 2); per component AC refinement to 1; DC refinement; per component AC refinement to 0.  Every scan that uses
    Huffman coding is preceded by its own optimal table(s).  Operations are collected symbolically first (('sym', table,
    symbol) | ('bits', value, count) | ('rst', n)) so that the tables can be built from their statistics."""
    n_comp = len(comps)

    def dc_first(units, al):
        ops, pred = [], {}
        for n, u in enumerate(units):
            if restart and n and n % restart == 0:
                ops.append(("rst", (n // restart - 1) % 8))
                pred = {}
            for ci, zz in u:
                v = int(zz[0]) >> al                                     # arithmetic shift: the DC point transform
                diff = v - pred.get(ci, 0)
                pred[ci] = v
                size = abs(diff).bit_length()
                ops.append(("sym", ("dc", comps[ci]["td"]), size))
                if size:
                    ops.append(("bits", diff if diff >= 0 else diff + (1 << size) - 1, size))
        return ops

    def dc_refine(units, al):
        ops = []
        for n, u in enumerate(units):
            if restart and n and n % restart == 0:
                ops.append(("rst", (n // restart - 1) % 8))
            for ci, zz in u:
                ops.append(("bits", (int(zz[0]) >> al) & 1, 1))
        return ops

    def eob_flush(ops, state):
        if state["eobrun"] > 0:
            nbits = state["eobrun"].bit_length() - 1
            ops.append(("sym", ("ac", 0), nbits << 4))
            if nbits:
                ops.append(("bits", state["eobrun"] & ((1 << nbits) - 1), nbits))
            state["eobrun"] = 0
        for b in state["be"]:
            ops.append(("bits", b, 1))
        state["be"] = []

    def ac_first(units, ss, se, al):
        ops, state = [], {"eobrun": 0, "be": []}
        for n, u in enumerate(units):
            if restart and n and n % restart == 0:
                eob_flush(ops, state)
                ops.append(("rst", (n // restart - 1) % 8))
            (ci, zz), = u
            r = 0
            for k in range(ss, se + 1):
                v = int(zz[k])
                t = abs(v) >> al                                         # the AC point transform truncates towards zero
                if t == 0:
                    r += 1
                    continue
                eob_flush(ops, state)
                while r > 15:
                    ops.append(("sym", ("ac", 0), 0xF0))
                    r -= 16
                size = t.bit_length()
                ops.append(("sym", ("ac", 0), (r << 4) | size))
                ops.append(("bits", t if v >= 0 else ((1 << size) - 1) ^ t, size))
                r = 0
            if r > 0:
                state["eobrun"] += 1
                if state["eobrun"] == 0x7FFF:
                    eob_flush(ops, state)
        eob_flush(ops, state)
        return ops

    def ac_refine(units, ss, se, al):
        ops, state = [], {"eobrun": 0, "be": []}
        for n, u in enumerate(units):
            if restart and n and n % restart == 0:
                eob_flush(ops, state)
                ops.append(("rst", (n // restart - 1) % 8))
            (ci, zz), = u
            absval = {k: abs(int(zz[k])) >> al for k in range(ss, se + 1)}
            eob = max([k for k in absval if absval[k] == 1], default=0)   # last coefficient that becomes non-zero in this scan
            r, br = 0, []
            for k in range(ss, se + 1):
                t = absval[k]
                if t == 0:
                    r += 1
                    continue
                while r > 15 and k <= eob:                               # a run of sixteen zeros that cannot be folded into an end-of-band
                    eob_flush(ops, state)
                    ops.append(("sym", ("ac", 0), 0xF0))
                    r -= 16
                    ops.extend(("bits", b, 1) for b in br)
                    br = []
                if t > 1:                                                # non-zero before this scan: one correction bit
                    br.append(t & 1)
                    continue
                eob_flush(ops, state)
                ops.append(("sym", ("ac", 0), (r << 4) | 1))
                ops.append(("bits", 0 if int(zz[k]) < 0 else 1, 1))
                ops.extend(("bits", b, 1) for b in br)
                br = []
                r = 0
            if r > 0 or br:
                state["eobrun"] += 1
                state["be"] += br
                if state["eobrun"] == 0x7FFF or len(state["be"]) > 900:
                    eob_flush(ops, state)
        eob_flush(ops, state)
        return ops

    all_ids = list(range(n_comp))
    script = [(all_ids, 0, 0, 0, 1, dc_first(scan_blocks(all_ids), 1))]
    for ci in all_ids:
        script.append(([ci], 1, 5, 0, 2, ac_first(scan_blocks([ci]), 1, 5, 2)))
    for ci in all_ids:
        script.append(([ci], 6, 63, 0, 2, ac_first(scan_blocks([ci]), 6, 63, 2)))
    for ci in all_ids:
        script.append(([ci], 1, 63, 2, 1, ac_refine(scan_blocks([ci]), 1, 63, 1)))
    script.append((all_ids, 0, 0, 1, 0, dc_refine(scan_blocks(all_ids), 0)))
    for ci in all_ids:
        script.append(([ci], 1, 63, 1, 0, ac_refine(scan_blocks([ci]), 1, 63, 0)))

    def seg(marker, body):
        return bytes([0xFF, marker]) + struct.pack(">H", len(body) + 2) + body

    out = bytearray(b"\xFF\xD8")
    out += seg(0xE0, b"JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    ntab = 1 if n_comp == 1 else 2
    out += seg(0xDB, b"".join(bytes([t]) + bytes(int(v) for v in qtabs[t].reshape(64)[_JPEG_ZIGZAG]) for t in range(ntab)))
    ids = [ord(c) for c in "RGB"] if rgb_ids else [1, 2, 3]
    sof = struct.pack(">BHHB", 8, H, W, n_comp)
    for ci, c in enumerate(comps):
        sof += bytes([ids[ci], (c["h"] << 4) | c["v"], c["tq"]])
    out += seg(0xC2, sof)
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))
    for comp_ids, ss, se, ah, al, ops in script:
        freq = {}
        for op in ops:
            if op[0] == "sym":
                f = freq.setdefault(op[1], {})
                f[op[2]] = f.get(op[2], 0) + 1
        tables = {key: _jpeg_huffman_table(f) for key, f in freq.items()}
        if tables:
            dht = b""
            for (kind, t), (counts, order, _) in sorted(tables.items()):
                dht += bytes([(0x10 if kind == "ac" else 0x00) | t]) + bytes(counts) + bytes(order)
            out += seg(0xC4, dht)
        sos = bytes([len(comp_ids)])
        for ci in comp_ids:
            sos += bytes([ids[ci], (comps[ci]["td"] << 4) | 0])         # DC table by component, AC table 0 (redefined per scan)
        out += seg(0xDA, sos + bytes([ss, se, (ah << 4) | al]))
        w = _JpegBits()
        for op in ops:
            if op[0] == "sym":
                code, length = tables[op[1]][2][op[2]]
                w.put(code, length)
            elif op[0] == "bits":
                w.put(op[1], op[2])
            else:
                w.marker(0xD0 + op[1])
        w.pad()
        out += w.data
    out += b"\xFF\xD9"
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_jpeg(path: str, img: np.ndarray, sampling: Tuple[int, int] = (1, 1), restart: int = 0, interleaved: bool = True,
               rgb_ids: bool = False, q_step: Tuple[int, int] = (2, 3), progressive: bool = False) -> None:
    """Baseline (SOF0) or, with `progressive`, progressive (SOF2: spectral selection + successive approximation, ten
    scans for a colour image) JPEG with its own optimal Huffman tables.  img: [h, w] grey or [h, w, 3] RGB.  `sampling` = luma
    (h, v) factors against 1x1 chroma: (1,1) 4:4:4, (2,1) 4:2:2, (1,2) 4:4:0, (2,2) 4:2:0, (4,1) 4:1:1.  `restart`:
    restart interval in MCUs (0 = none).  `interleaved` False writes one scan per component.  `rgb_ids`: store R, G, B
    themselves under the component ids 'R', 'G', 'B' (no colour transform on either side)."""
    img = np.asarray(img, dtype=np.uint8)
    grey = img.ndim == 2 or img.shape[2] == 1
    H, W = img.shape[:2]
    if grey:
        planes = [img.reshape(H, W).astype(np.float64)]
        factors = [(1, 1)]
    else:
        rgb = img[:, :, :3].astype(np.float64)
        if rgb_ids:
            planes = [rgb[:, :, 0], rgb[:, :, 1], rgb[:, :, 2]]
        else:
            r, g, b = rgb[:, :, 0], rgb[:, :, 1], rgb[:, :, 2]
            planes = [0.299 * r + 0.587 * g + 0.114 * b,
                      128.0 - 0.168736 * r - 0.331264 * g + 0.5 * b,
                      128.0 + 0.5 * r - 0.418688 * g - 0.081312 * b]
        factors = [tuple(sampling), (1, 1), (1, 1)]
    hmax = max(f[0] for f in factors)
    vmax = max(f[1] for f in factors)
    mcu_w, mcu_h = 8 * hmax, 8 * vmax
    mcus_x, mcus_y = (W + mcu_w - 1) // mcu_w, (H + mcu_h - 1) // mcu_h
    # DCT-II basis
    k = np.arange(8)
    C = np.sqrt(2.0 / 8.0) * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16.0)
    C[0, :] = np.sqrt(1.0 / 8.0)
    ii, jj = np.mgrid[0:8, 0:8]
    qtabs = [np.clip(2 + (ii + jj) * q_step[0], 1, 255).astype(np.int64), np.clip(3 + (ii + jj) * q_step[1], 1, 255).astype(np.int64)]
    comps = []
    for ci, (plane, (fh, fv)) in enumerate(zip(planes, factors)):
        full = np.pad(plane, ((0, mcus_y * mcu_h - H), (0, mcus_x * mcu_w - W)), mode="edge")
        sh, sv = hmax // fh, vmax // fv                                 # down-sampling of this component
        sub = full.reshape(full.shape[0] // sv, sv, full.shape[1] // sh, sh).mean(axis=(1, 3))
        q = qtabs[0 if ci == 0 else 1]
        by, bx = sub.shape[0] // 8, sub.shape[1] // 8
        coef = np.zeros((by, bx, 64), dtype=np.int64)
        for y in range(by):
            for x in range(bx):
                blk = sub[8 * y:8 * y + 8, 8 * x:8 * x + 8] - 128.0
                d = C @ blk @ C.T
                coef[y, x] = np.rint(d / q).astype(np.int64).reshape(64)[_JPEG_ZIGZAG]
        comps.append(dict(h=fh, v=fv, coef=coef, tq=0 if ci == 0 else 1, td=0 if ci == 0 else 1,
                          real_bx=((W * fh + hmax - 1) // hmax + 7) // 8, real_by=((H * fv + vmax - 1) // vmax + 7) // 8))

    # ---- scans as lists of (component, block) in coding order, with restart boundaries
    def scan_blocks(component_ids):
        units = []                                                      # one entry per MCU: [(ci, zigzag coefficients), ...]
        if len(component_ids) == 1:
            c = comps[component_ids[0]]
            for y in range(c["real_by"]):
                for x in range(c["real_bx"]):
                    units.append([(component_ids[0], c["coef"][y, x])])
        else:
            for my in range(mcus_y):
                for mx in range(mcus_x):
                    u = []
                    for ci in component_ids:
                        c = comps[ci]
                        for y in range(c["v"]):
                            for x in range(c["h"]):
                                u.append((ci, c["coef"][my * c["v"] + y, mx * c["h"] + x]))
                    units.append(u)
        return units

    def symbols_of(units):
        """[(kind, table, symbol, extra bits value, extra bits count) | ('rst', n)]"""
        out = []
        pred = {}
        for n, u in enumerate(units):
            if restart and n and n % restart == 0:
                out.append(("rst", (n // restart - 1) % 8))
                pred = {}
            for ci, zz in u:
                diff = int(zz[0]) - pred.get(ci, 0)
                pred[ci] = int(zz[0])
                size = abs(diff).bit_length()
                out.append(("dc", comps[ci]["td"], size, diff if diff >= 0 else diff + (1 << size) - 1, size))
                run = 0
                last = max([i for i in range(1, 64) if zz[i] != 0], default=0)
                for i in range(1, last + 1):
                    v = int(zz[i])
                    if v == 0:
                        run += 1
                        continue
                    while run > 15:
                        out.append(("ac", comps[ci]["td"], 0xF0, 0, 0))
                        run -= 16
                    size = abs(v).bit_length()
                    out.append(("ac", comps[ci]["td"], (run << 4) | size, v if v >= 0 else v + (1 << size) - 1, size))
                    run = 0
                if last < 63:
                    out.append(("ac", comps[ci]["td"], 0x00, 0, 0))
        return out

    if progressive:
        _write_progressive_jpeg(path, H, W, comps, qtabs, mcus_x, mcus_y, restart, rgb_ids, scan_blocks)
        return
    scans = [list(range(len(comps)))] if (interleaved or len(comps) == 1) else [[ci] for ci in range(len(comps))]
    scan_syms = [symbols_of(scan_blocks(ids)) for ids in scans]
    freq = {("dc", 0): {}, ("dc", 1): {}, ("ac", 0): {}, ("ac", 1): {}}
    for syms in scan_syms:
        for s_ in syms:
            if s_[0] != "rst":
                f = freq[(s_[0], s_[1])]
                f[s_[2]] = f.get(s_[2], 0) + 1
    tables = {key: _jpeg_huffman_table(f) for key, f in freq.items() if f}

    def seg(marker, body):
        return bytes([0xFF, marker]) + struct.pack(">H", len(body) + 2) + body

    out = bytearray(b"\xFF\xD8")
    out += seg(0xE0, b"JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    out += seg(0xFE, b"par_raytracer_amd test texture")
    ntab = 1 if len(comps) == 1 else 2
    out += seg(0xDB, b"".join(bytes([t]) + bytes(int(v) for v in qtabs[t].reshape(64)[_JPEG_ZIGZAG]) for t in range(ntab)))
    ids = [ord(c) for c in "RGB"] if rgb_ids else [1, 2, 3]
    sof = struct.pack(">BHHB", 8, H, W, len(comps))
    for ci, c in enumerate(comps):
        sof += bytes([ids[ci], (c["h"] << 4) | c["v"], c["tq"]])
    out += seg(0xC0, sof)
    dht = b""
    for (kind, t), (counts, order, _) in sorted(tables.items()):
        dht += bytes([(0x10 if kind == "ac" else 0x00) | t]) + bytes(counts) + bytes(order)
    out += seg(0xC4, dht)
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))
    for comp_ids, syms in zip(scans, scan_syms):
        sos = bytes([len(comp_ids)])
        for ci in comp_ids:
            sos += bytes([ids[ci], (comps[ci]["td"] << 4) | comps[ci]["td"]])
        out += seg(0xDA, sos + b"\x00\x3F\x00")
        acc, nacc = 0, 0
        data = bytearray()

        def flush_bits():
            nonlocal acc, nacc
            while nacc >= 8:
                byte = (acc >> (nacc - 8)) & 0xFF
                data.append(byte)
                if byte == 0xFF:
                    data.append(0)
                nacc -= 8
            acc &= (1 << nacc) - 1

        def put(value, nbits):
            nonlocal acc, nacc
            acc = (acc << nbits) | (value & ((1 << nbits) - 1))
            nacc += nbits
            flush_bits()

        def pad():
            if nacc % 8:
                put((1 << (8 - nacc % 8)) - 1, 8 - nacc % 8)

        for s_ in syms:
            if s_[0] == "rst":
                pad()
                data += bytes([0xFF, 0xD0 + s_[1]])
                continue
            code, length = tables[(s_[0], s_[1])][2][s_[2]]
            put(code, length)
            if s_[4]:
                put(s_[3], s_[4])
        pad()
        out += data
    out += b"\xFF\xD9"
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_texture(path: str, img: np.ndarray, encoding: str) -> None:
    enc = {
        "png": lambda: write_png(path, img),
        "png16": lambda: write_png(path, img, sixteen_bit=True),
        "png_palette": lambda: write_png(path, img, palette=True),
        "png_palette_alpha": lambda: write_png(path, img, palette=True, palette_alpha=True),
        "png_i": lambda: write_png(path, img, interlace=True),                                  # Adam7
        "png16_i": lambda: write_png(path, img, sixteen_bit=True, interlace=True),
        "png_g1": lambda: write_png(path, img, bits=1),                                         # grey, values 0..1
        "png_g2": lambda: write_png(path, img, bits=2),
        "png_g4_i": lambda: write_png(path, img, bits=4, interlace=True),
        "png_p4": lambda: write_png(path, img, palette=True, bits=4),                           # <= 16 colours
        "png_p1_alpha_i": lambda: write_png(path, img, palette=True, palette_alpha=True, bits=1, interlace=True),
        "png_key": lambda: write_png(path, img, key=np.asarray(img).reshape(-1, 1 if np.asarray(img).ndim == 2 else np.asarray(img).shape[2])[0]),
        "png16_key_i": lambda: write_png(path, img, sixteen_bit=True, interlace=True,
                                         key=np.asarray(img).reshape(-1, 1 if np.asarray(img).ndim == 2 else np.asarray(img).shape[2])[0]),
        "png_g2_key": lambda: write_png(path, img, bits=2, key=[int(np.asarray(img).reshape(-1)[0])]),
        "tga": lambda: write_tga(path, img),
        "tga_top": lambda: write_tga(path, img, top_down=True),
        "tga_rle": lambda: write_tga(path, img, rle=True),
        "tga16": lambda: write_tga(path, img, kind="16"),
        "tga16_rle": lambda: write_tga(path, img, kind="16", rle=True, top_down=True),
        "tga_ga": lambda: write_tga(path, img, kind="ga"),
        "tga_map24": lambda: write_tga(path, img, kind="map24"),
        "tga_map32_rle": lambda: write_tga(path, img, kind="map32", rle=True),
        "tga_map16": lambda: write_tga(path, img, kind="map16", top_down=True),
        "tga_map24_i16": lambda: write_tga(path, img, kind="map24_i16"),
        "bmp": lambda: write_bmp(path, img),
        "bmp_top": lambda: write_bmp(path, img, "24_top"),
        "bmp_os2": lambda: write_bmp(path, img, "os2_24"),
        "bmp_os2_8": lambda: write_bmp(path, img, "os2_8"),
        "bmp8": lambda: write_bmp(path, img, "8"),
        "bmp4": lambda: write_bmp(path, img, "4"),
        "bmp16": lambda: write_bmp(path, img, "16"),
        "bmp16_565": lambda: write_bmp(path, img, "16_565"),
        "bmp32": lambda: write_bmp(path, img, "32"),
        "bmp32_v4": lambda: write_bmp(path, img, "32_v4"),
        "pnm": lambda: write_pnm(path, img),
        "pic": lambda: write_pic(path, img),
        "pic_raw": lambda: write_pic(path, img, "raw"),
        "pic_pure": lambda: write_pic(path, img, "pure"),
        "hdr": lambda: write_hdr(path, img),
        "hdr_flat": lambda: write_hdr(path, img, rle=False, magic="#?RGBE"),
        "psd": lambda: write_psd(path, img),
        "psd_rle": lambda: write_psd(path, img, rle=True),
        "psd16": lambda: write_psd(path, img, sixteen_bit=True),
        "gif": lambda: write_gif(path, img),
        "gif_i": lambda: write_gif(path, img, interlace=True),
        "gif_t": lambda: write_gif(path, img, transparent=True),
        "gif_local_i_t": lambda: write_gif(path, img, interlace=True, transparent=True, local_table=True),
        "gif_canvas": lambda: write_gif(path, img, canvas=(np.asarray(img).shape[1] + 7, np.asarray(img).shape[0] + 5, 3, 2), bgindex=1),
        "jpg": lambda: write_jpeg(path, img),                                             # 4:4:4 (or grey)
        "jpg422": lambda: write_jpeg(path, img, sampling=(2, 1)),
        "jpg440": lambda: write_jpeg(path, img, sampling=(1, 2)),
        "jpg420": lambda: write_jpeg(path, img, sampling=(2, 2)),
        "jpg411": lambda: write_jpeg(path, img, sampling=(4, 1)),
        "jpg420_rst": lambda: write_jpeg(path, img, sampling=(2, 2), restart=3),
        "jpg_scans": lambda: write_jpeg(path, img, sampling=(2, 2), interleaved=False, restart=5),
        "jpg_rgb": lambda: write_jpeg(path, img, rgb_ids=True),
        "jpg_prog": lambda: write_jpeg(path, img, progressive=True),                      # progressive, 4:4:4 (or grey)
        "jpg_prog420": lambda: write_jpeg(path, img, sampling=(2, 2), progressive=True),
        "jpg_prog422_rst": lambda: write_jpeg(path, img, sampling=(2, 1), progressive=True, restart=4),
    }
    enc[encoding]()


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


# ----------------------------------------------------------------------------------------
# N1: textured gallery - every texture slot, every image encoding the loader decodes
# ----------------------------------------------------------------------------------------

def _quad(pos, nrm, uv, corners, normal, uvs):
    """Append a quad (4 corners, counter-clockwise seen from `normal`) -> two triangles of vertex ids."""
    base = len(pos)
    for c, t in zip(corners, uvs):
        pos.append(list(c)); nrm.append(list(normal)); uv.append(list(t))
    return [[base, base + 1, base + 2], [base, base + 2, base + 3]]


def textured_gallery(sphere_segments: int = 20, sphere_rings: int = 10) -> ObjScene:
    """Row N1 of SURVEY.md 8f: ambient / diffuse / specular / alpha / bump maps (raytracer.cpp:439-502, 547-552).

    A floor (tiled checker + bump, texture coordinates running negative and past 1), a lit right wall (RGBA
    diffuse + specular map), a back wall (2x2 ambient map: the `size - 2` scale is 0 there), a translucent fence
    whose alpha map has holes (alpha <= 0.05 passes the ray through with its bounce budget intact), a bumpy ball
    and a panel with a palette texture.  Every image encoding image_in.cpp decodes appears once."""
    rng = np.random.default_rng(20240611)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    # floor: y = 0, normal +y, uv tiled 3x with a negative origin
    add("floor", _quad(pos, nrm, uv, [(-3, 0, 3), (3, 0, 3), (3, 0, -3), (-3, 0, -3)], (0, 1, 0),
                       [(-1.25, -0.5), (1.75, -0.5), (1.75, 2.5), (-1.25, 2.5)]), "floor")
    # right wall: x = 3, normal -x (lit by the default light)
    add("right_wall", _quad(pos, nrm, uv, [(3, 0, 3), (3, 3, 3), (3, 3, -3), (3, 0, -3)], (-1, 0, 0),
                            [(0, 0), (0, 1), (2, 1), (2, 0)]), "right_wall")
    # left wall: x = -3, normal +x, untextured material
    add("left_wall", _quad(pos, nrm, uv, [(-3, 0, -3), (-3, 3, -3), (-3, 3, 3), (-3, 0, 3)], (1, 0, 0),
                           [(0, 0), (0, 1), (1, 1), (1, 0)]), "plain")
    # back wall: z = -3, normal +z
    add("back_wall", _quad(pos, nrm, uv, [(-3, 0, -3), (3, 0, -3), (3, 3, -3), (-3, 3, -3)], (0, 0, 1),
                           [(0, 0), (1, 0), (1, 1), (0, 1)]), "back_wall")
    # fence: z = 0.75, normal +z (towards the camera), alpha-mapped
    add("fence", _quad(pos, nrm, uv, [(-1.6, 0, 0.75), (1.6, 0, 0.75), (1.6, 1.6, 0.75), (-1.6, 1.6, 0.75)], (0, 0, 1),
                       [(0, 0), (2, 0), (2, 1), (0, 1)]), "fence")
    # tilted panel on the left
    add("panel", _quad(pos, nrm, uv, [(-2.6, 0.4, 0.2), (-1.2, 0.4, -0.9), (-1.2, 1.9, -0.9), (-2.6, 1.9, 0.2)],
                       tuple(_normalize(np.array([[1.1, 0.0, 1.4]]))[0]),
                       [(0, 0), (1, 0), (1, 1), (0, 1)]), "panel")
    # ball behind the fence
    c = np.array([0.5, 0.8, -0.9]); radius = 0.8
    base = len(pos)
    pos.append(list(c + [0, radius, 0])); nrm.append([0, 1, 0]); uv.append([0.5, 0.0])
    for r in range(1, sphere_rings):
        th = np.pi * r / sphere_rings
        for sg in range(sphere_segments + 1):                      # seam duplicated so u runs 0..2 without a jump
            ph = 2 * np.pi * sg / sphere_segments
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            pos.append(list(c + radius * n)); nrm.append(list(n)); uv.append([2.0 * sg / sphere_segments, r / sphere_rings])    # v grows downwards: positive uv area, so tangents exist (mesh.h:93)
    pos.append(list(c - [0, radius, 0])); nrm.append([0, -1, 0]); uv.append([0.5, 1.0])
    south = len(pos) - 1

    def ring(r, sg):
        return base + 1 + (r - 1) * (sphere_segments + 1) + sg

    tris = []
    for sg in range(sphere_segments):
        tris.append([base, ring(1, sg + 1), ring(1, sg)])
    for r in range(1, sphere_rings - 1):
        for sg in range(sphere_segments):
            a, b = ring(r, sg), ring(r, sg + 1)
            d, e = ring(r + 1, sg), ring(r + 1, sg + 1)
            tris.append([a, b, e])
            tris.append([a, e, d])
    for sg in range(sphere_segments):
        tris.append([south, ring(sphere_rings - 1, sg), ring(sphere_rings - 1, sg + 1)])
    add("ball", tris, "ball")

    # ---- images -------------------------------------------------------------------------------------------
    def checker(h, w, cell, c0, c1):
        yy, xx = np.mgrid[0:h, 0:w]
        m = ((yy // cell + xx // cell) % 2).astype(bool)
        img = np.where(m[:, :, None], np.array(c1, dtype=np.uint8), np.array(c0, dtype=np.uint8))
        return (img.astype(np.int32) + rng.integers(-12, 13, size=img.shape)).clip(0, 255).astype(np.uint8)

    def waves(h, w, fx, fy):
        yy, xx = np.mgrid[0:h, 0:w]
        v = 0.5 + 0.5 * np.sin(2 * np.pi * fx * xx / w) * np.cos(2 * np.pi * fy * yy / h)
        return (v * 255).round().astype(np.uint8)

    textures = {}
    textures["floor_kd.png"] = (checker(64, 48, 8, (40, 60, 200), (230, 220, 190)), "png")                  # RGB, all 5 filters
    textures["floor_bump.tga"] = (waves(48, 48, 3, 2), "tga_rle")                                             # grey, run-length
    rgba = np.concatenate([checker(37, 29, 5, (220, 40, 40), (250, 240, 120)),
                           rng.integers(0, 256, size=(37, 29, 1), dtype=np.uint8)], axis=2)
    textures["wall_kd.png"] = (rgba, "png")                                                                   # RGBA
    textures["wall_ks.ppm"] = (checker(32, 32, 4, (20, 20, 20), (255, 255, 255)), "pnm")                      # binary PPM
    textures["back_ka.bmp"] = (np.array([[[255, 80, 20], [30, 200, 90]], [[10, 40, 250], [240, 240, 60]]], dtype=np.uint8), "bmp")   # 2x2
    textures["back_kd.tga"] = (checker(40, 56, 7, (90, 160, 90), (200, 230, 200)), "tga")                     # BGR, bottom-up
    holes = waves(64, 64, 4, 4)
    holes[holes < 90] = 0                                                                                     # holes: alpha 0
    textures["fence_d.pgm"] = (holes, "pnm")                                                                  # binary PGM
    textures["fence_kd.png"] = (checker(32, 64, 4, (120, 90, 40), (180, 140, 70)), "png16")                   # 16-bit samples
    pal = checker(48, 48, 6, (30, 30, 30), (240, 200, 40)) // 32 * 32                                         # few distinct colours
    textures["panel_kd.png"] = (pal, "png_palette")
    pal_a = np.concatenate([pal, (waves(48, 48, 2, 1) // 64 * 64)[:, :, None]], axis=2).astype(np.uint8)
    textures["panel_ka.png"] = (pal_a, "png_palette_alpha")                                                    # palette + tRNS -> RGBA
    textures["panel_d.tga"] = ((128 + waves(33, 31, 1, 2) // 2).astype(np.uint8), "tga_top")                  # grey, top-down
    textures["ball_kd.tga"] = (np.concatenate([checker(32, 64, 8, (200, 200, 220), (60, 60, 160)),
                                               np.full((32, 64, 1), 255, dtype=np.uint8)], axis=2), "tga_rle")   # BGRA, run-length
    textures["ball_bump.png"] = (rng.integers(60, 200, size=(32, 32), dtype=np.uint8), "png")                 # grey PNG
    textures["ball_ks.png"] = (rng.integers(0, 256, size=(16, 24, 2), dtype=np.uint8), "png")                 # grey + alpha: 2 channels -> (r, g, 0)

    materials = [
        MtlMaterial("floor", Ns=30.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.3, 0.3, 0.3), map_Kd="floor_kd.png", map_bump="floor_bump.tga"),
        MtlMaterial("right_wall", Ns=60.0, Ka=(0.5, 0.5, 0.5), Kd=(0.8, 0.8, 0.8), Ks=(0.1, 0.1, 0.1), map_Kd="wall_kd.png", map_Ks="wall_ks.ppm"),
        MtlMaterial("plain", Ka=(0.4, 0.7, 0.4), Kd=(0.4, 0.7, 0.4), Ks=(0.2, 0.2, 0.2)),
        MtlMaterial("back_wall", Ka=(1.0, 1.0, 1.0), Kd=(0.7, 0.7, 0.7), Ks=(0.0, 0.0, 0.0), map_Ka="back_ka.bmp", map_Kd="back_kd.tga"),
        MtlMaterial("fence", Ns=5.0, d=0.9, Ka=(0.5, 0.4, 0.3), Kd=(1.0, 1.0, 1.0), Ks=(0.05, 0.05, 0.05), map_Kd="fence_kd.png", map_d="fence_d.pgm"),
        MtlMaterial("panel", Ns=15.0, Ka=(0.8, 0.8, 0.8), Kd=(1.0, 1.0, 1.0), Ks=(0.4, 0.4, 0.4), map_Ka="panel_ka.png", map_Kd="panel_kd.png", map_d="panel_d.tga"),
        MtlMaterial("ball", Ns=40.0, Ka=(0.5, 0.5, 0.6), Kd=(0.9, 0.9, 1.0), Ks=(0.6, 0.6, 0.6), map_Kd="ball_kd.tga", map_Ks="ball_ks.png", map_bump="ball_bump.png"),
    ]
    return ObjScene(
        name="textured_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.3, 1.4, 4.6), camera_facing=(-0.05, -0.18, -1.0), fov=60.0)


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


def jpeg_gallery() -> ObjScene:
    """JPEG textures through the loader and the texture path: thirteen panels in two rows over a floor, one per JPEG layout
    write_jpeg produces - baseline grey, 4:4:4, 4:2:2, 4:4:0, 4:2:0, 4:1:1, 4:2:0 with a restart interval, one scan per
    component with a restart interval, RGB component ids; progressive 4:4:4, 4:2:0, 4:2:2 with a restart interval, grey -
    with sizes that are not multiples of the MCU, plus grey JPEGs as bump and alpha maps.  The decoded bytes depend on the decoder's inverse DCT, upsampling filter and colour arithmetic,
    which is what the fixture pins against the reference's decoder."""
    rng = np.random.default_rng(20241004)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def picture(h, w, kind):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(2 * np.pi * (xx / w * 1.5 + kind * 0.13)) * np.cos(2 * np.pi * yy / h),
                         128 + 110 * np.cos(2 * np.pi * (yy / h * 2.0 + kind * 0.07)),
                         ((xx // 5 + yy // 7 + kind) % 2) * 200 + 30], axis=2)
        img = base + rng.integers(-25, 26, size=base.shape)
        img[h // 3:h // 3 + 4, :, :] = (255, 0, 0) if kind % 2 else (0, 0, 255)      # saturated edges: clamping in the colour arithmetic
        img[:, w // 2:w // 2 + 2, :] = 255
        return img.clip(0, 255).astype(np.uint8)

    layouts = [("grey", "jpg", (23, 41)), ("c444", "jpg", (30, 37)), ("c422", "jpg422", (33, 47)), ("c440", "jpg440", (45, 26)),
               ("c420", "jpg420", (50, 61)), ("c411", "jpg411", (19, 70)), ("c420r", "jpg420_rst", (64, 48)),
               ("scans", "jpg_scans", (35, 52)), ("rgbid", "jpg_rgb", (17, 24)),
               ("prog", "jpg_prog", (29, 43)), ("prog420", "jpg_prog420", (54, 39)), ("prog422r", "jpg_prog422_rst", (31, 58)),
               ("proggrey", "jpg_prog", (37, 21))]
    textures = {}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.jpg",
                             map_bump="floor_bump.jpg")]
    textures["floor_kd.jpg"] = (picture(72, 96, 11), "jpg420")
    yy, xx = np.mgrid[0:40, 0:56]
    textures["floor_bump.jpg"] = ((127.5 + 120 * np.sin(xx * 0.7) * np.cos(yy * 0.5)).astype(np.uint8), "jpg")       # grey height map
    for k, (name, enc, (h, w)) in enumerate(layouts):
        col, row = k % 7, k // 7
        x0, y0 = -4.15 + col * 1.2, 0.15 + row * 1.3
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 1.05, y0, z), (x0 + 1.05, y0 + 1.05, z), (x0, y0 + 1.05, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        img = picture(h, w, k)
        textures[name + ".jpg"] = (img[:, :, 1] if name in ("grey", "proggrey") else img, enc)
        kw = {}
        if name == "c444":                                              # a grey JPEG as alpha map: holes in the panel
            a = (128 + 127 * np.sin(xx[:32, :32] * 0.9) * np.sin(yy[:32, :32] * 0.8))
            a[a < 100] = 0
            textures["c444_d.jpg"] = (a.astype(np.uint8), "jpg")
            kw["map_d"] = "c444_d.jpg"
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15),
                                     map_Kd=name + ".jpg", **kw))
    return ObjScene(
        name="jpeg_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def png_gallery() -> ObjScene:
    """The PNG corners of the loader through the texture path: a floor and twelve panels whose diffuse maps are Adam7
    interlaced (8- and 16-bit), 1 / 2 / 4-bit grey, 4-bit and 1-bit palettes (the latter with tRNS, interlaced), and grey /
    RGB images with a colour-key tRNS chunk (which adds an alpha channel: 2 and 4 channels), at sizes that leave some
    interlace passes empty."""
    rng = np.random.default_rng(20241005)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def blobs(h, w, levels):
        yy, xx = np.mgrid[0:h, 0:w]
        v = 0.5 + 0.5 * np.sin(xx * 0.9 + 0.3) * np.cos(yy * 0.7) + rng.uniform(-0.15, 0.15, size=(h, w))
        return np.clip((v * levels).astype(np.int64), 0, levels - 1).astype(np.uint8)

    def colour(h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([40 + 200 * ((xx // 3 + yy // 2) % 2), 128 + 100 * np.sin(yy * 0.8), 30 + 7 * xx], axis=2) + rng.integers(-20, 21, size=(h, w, 3))
        return img.clip(0, 255).astype(np.uint8)

    def keyed(img):                                                     # make the key colour (pixel 0) occur in patches
        out = img.copy()
        out[::3, ::2] = img.reshape(-1, img.shape[-1] if img.ndim == 3 else 1)[0] if img.ndim == 3 else img.reshape(-1)[0]
        return out

    pal4 = (blobs(21, 30, 4)[:, :, None] * np.array([60, 30, 80], dtype=np.uint8) + np.array([10, 90, 0], dtype=np.uint8)).astype(np.uint8)
    pal1 = np.where(blobs(13, 9, 2)[:, :, None] > 0, np.array([240, 200, 30, 255], dtype=np.uint8), np.array([30, 40, 200, 90], dtype=np.uint8)).astype(np.uint8)
    layouts = [
        ("i_rgb", colour(29, 23), "png_i"), ("i_rgba16", np.concatenate([colour(17, 35), rng.integers(120, 256, size=(17, 35, 1), dtype=np.uint8)], axis=2), "png16_i"),
        ("g1", blobs(19, 27, 2), "png_g1"), ("g2", blobs(22, 13, 4), "png_g2"), ("g4_i", blobs(15, 31, 16), "png_g4_i"),
        ("p4", pal4, "png_p4"), ("p1_alpha_i", pal1, "png_p1_alpha_i"),
        ("key_grey", keyed(blobs(16, 20, 256)), "png_key"), ("key_rgb", keyed(colour(18, 26)), "png_key"),
        ("key_rgb16_i", keyed(colour(11, 14)), "png16_key_i"), ("key_g2", keyed(blobs(9, 21, 4)), "png_g2_key"),
        ("i_thin", colour(37, 2), "png_i"),                          # (a 1-pixel-wide texture makes the reference read out of bounds: its size - 2 scale wraps)
    ]
    textures = {"floor_kd.png": (colour(40, 64), "png_i")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.png")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 6, k // 6
        x0, y0 = -3.9 + col * 1.3, 0.15 + row * 1.4
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 1.15, y0, z), (x0 + 1.15, y0 + 1.15, z), (x0, y0 + 1.15, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".png"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".png"))
    return ObjScene(
        name="png_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def bmp_gallery() -> ObjScene:
    """Every BMP flavour the reference's decoder accepts as diffuse maps: 24-bit bottom-up and top-down, OS/2 headers
    (24-bit and 8-bit palette), 8- and 4-bit palettes, 16-bit 5-5-5, 16-bit 5-6-5 with BITFIELDS masks (which that decoder
    reads twelve bytes late), plain 32-bit with a real alpha channel and with an all-zero one, and a 108-byte header with
    masks in A, R, G, B byte order."""
    rng = np.random.default_rng(20241006)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def colour(h, w, levels=256):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([40 + 200 * ((xx // 3 + yy // 2) % 2), 128 + 100 * np.sin(yy * 0.8), 30 + 7 * xx], axis=2) + rng.integers(-20, 21, size=(h, w, 3))
        img = img.clip(0, 255)
        if levels < 256:
            img = img // (256 // levels) * (256 // levels)
        return img.astype(np.uint8)

    def with_alpha(img, zero=False):
        a = np.zeros(img.shape[:2] + (1,), dtype=np.uint8) if zero else rng.integers(1, 256, size=img.shape[:2] + (1,), dtype=np.uint8)
        return np.concatenate([img, a], axis=2)

    layouts = [("b24", colour(21, 30), "bmp"), ("b24_top", colour(17, 23), "bmp_top"), ("os2_24", colour(13, 9), "bmp_os2"),
               ("os2_8", colour(19, 22, 4), "bmp_os2_8"), ("b8", colour(26, 31, 4), "bmp8"), ("b4", colour(15, 13, 2), "bmp4"),
               ("b16", colour(20, 27), "bmp16"), ("b16_565", colour(18, 25), "bmp16_565"), ("b32", with_alpha(colour(14, 19)), "bmp32"),
               ("b32_zero_alpha", with_alpha(colour(12, 17), zero=True), "bmp32"), ("b32_v4", with_alpha(colour(16, 11)), "bmp32_v4")]
    textures = {"floor_kd.bmp": (colour(40, 64), "bmp")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.bmp")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 6, k // 6
        x0, y0 = -3.9 + col * 1.3, 0.15 + row * 1.4
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 1.15, y0, z), (x0 + 1.15, y0 + 1.15, z), (x0, y0 + 1.15, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".bmp"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".bmp"))
    return ObjScene(
        name="bmp_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def tga_gallery() -> ObjScene:
    """The TGA corners of the loader as diffuse maps: 15 / 16-bit 5-5-5 pixels (raw, and run-length top-down), 16-bit grey +
    alpha (two channels), colour maps with 24-bit, 32-bit (run-length) and 15-bit entries, a colour map addressed by 16-bit
    indices - all with an image id, and the maps with a non-zero first-entry field."""
    rng = np.random.default_rng(20241007)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def colour(h, w, levels=256):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([40 + 200 * ((xx // 4 + yy // 3) % 2), 128 + 100 * np.sin(yy * 0.6), 30 + 6 * xx], axis=2) + rng.integers(-20, 21, size=(h, w, 3))
        img = img.clip(0, 255)
        if levels < 256:
            img = img // (256 // levels) * (256 // levels)
        return img.astype(np.uint8)

    few = colour(19, 24, 4)
    few_a = np.concatenate([few, ((few[:, :, :1].astype(np.int32) * 3 + 40) % 256).astype(np.uint8)], axis=2)
    many = colour(24, 30)                                               # > 256 colours: needs 16-bit indices
    layouts = [("t16", colour(21, 29), "tga16"), ("t16_rle", colour(18, 22, 8), "tga16_rle"),
               ("ga", rng.integers(0, 256, size=(15, 20, 2), dtype=np.uint8), "tga_ga"),
               ("map24", few, "tga_map24"), ("map32_rle", few_a, "tga_map32_rle"), ("map16", colour(17, 13, 4), "tga_map16"),
               ("map24_i16", many, "tga_map24_i16")]
    textures = {"floor_kd.tga": (colour(40, 64), "tga16")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.tga")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 4, k // 4
        x0, y0 = -3.4 + col * 1.75, 0.15 + row * 1.45
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 1.5, y0, z), (x0 + 1.5, y0 + 1.25, z), (x0, y0 + 1.25, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".tga"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".tga"))
    return ObjScene(
        name="tga_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def gif_gallery() -> ObjScene:
    """GIF diffuse maps the way the reference's decoder reads them (first image, always RGBA): plain, interlaced, with a
    transparent index (those pixels keep the background colour with alpha 0), a local colour table, an image placed on a
    larger logical screen, and a 256-colour image whose LZW stream fills and resets the code table."""
    rng = np.random.default_rng(20241008)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def colour(h, w, levels):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([40 + 200 * ((xx // 4 + yy // 3) % 2), 128 + 100 * np.sin(yy * 0.6), 30 + 6 * xx], axis=2)
        return (img.clip(0, 255) // (256 // levels) * (256 // levels)).astype(np.uint8)

    def keyed(img):
        out = img.copy()
        out[2::4, 1::3] = img[0, 0]
        return out

    noisy = rng.integers(0, 256, size=(48, 56, 1), dtype=np.uint8).repeat(3, axis=2)
    noisy[:, :, 1] = 255 - noisy[:, :, 0]
    layouts = [("plain", colour(25, 33, 4), "gif"), ("inter", colour(30, 21, 4), "gif_i"), ("transp", keyed(colour(19, 28, 4)), "gif_t"),
               ("local", keyed(colour(27, 18, 4)), "gif_local_i_t"), ("canvas", colour(16, 22, 4), "gif_canvas"), ("noisy", noisy, "gif_i")]
    textures = {"floor_kd.gif": (colour(40, 64, 8), "gif")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.gif")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 3, k // 3
        x0, y0 = -3.3 + col * 2.3, 0.15 + row * 1.45
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 2.0, y0, z), (x0 + 2.0, y0 + 1.25, z), (x0, y0 + 1.25, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".gif"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".gif"))
    return ObjScene(
        name="gif_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def psd_gallery() -> ObjScene:
    """Photoshop composites as diffuse maps: RGB raw, RGB PackBits, 16-bit, and RGBA (raw and PackBits) with every alpha
    value, so that the decoder's float un-blending from the white matte is exercised over its whole range."""
    rng = np.random.default_rng(20241009)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def colour(h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([40 + 200 * ((xx // 4 + yy // 3) % 2), 128 + 100 * np.sin(yy * 0.6), 30 + 6 * xx], axis=2) + rng.integers(-25, 26, size=(h, w, 3))
        img = img.clip(0, 255).astype(np.uint8)
        img[h // 2:h // 2 + 3] = img[h // 2, 0]                         # flat rows: runs for PackBits
        return img

    def with_alpha(img):
        h, w, _ = img.shape
        a = (np.arange(h * w).reshape(h, w) * 7 % 256).astype(np.uint8)  # every alpha value
        a[:2] = 255
        a[2:4] = 0
        return np.concatenate([img, a[:, :, None]], axis=2)

    layouts = [("rgb", colour(21, 30), "psd"), ("rgb_rle", colour(26, 19), "psd_rle"), ("rgb16", colour(14, 23), "psd16"),
               ("rgba", with_alpha(colour(24, 32)), "psd"), ("rgba_rle", with_alpha(colour(20, 27)), "psd_rle")]
    textures = {"floor_kd.psd": (colour(40, 64), "psd_rle")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.psd")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 3, k // 3
        x0, y0 = -3.3 + col * 2.3, 0.15 + row * 1.45
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 2.0, y0, z), (x0 + 2.0, y0 + 1.25, z), (x0, y0 + 1.25, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".psd"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".psd"))
    return ObjScene(
        name="psd_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def hdr_gallery() -> ObjScene:
    """Radiance .hdr files as diffuse maps, tone-mapped to 8 bits the way the reference's decoder does it: run-length
    scanlines, flat pixels in a wide file (the old format), a file narrower than 8 pixels (always flat), pixels with a zero
    exponent, radiances over several decades."""
    rng = np.random.default_rng(20241010)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def radiance(h, w, decades):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([0.5 + 0.5 * np.sin(xx * 0.5) * np.cos(yy * 0.4), 0.5 + 0.5 * np.cos(yy * 0.7), (xx + yy) / float(w + h)], axis=2)
        img = base * 10.0 ** rng.uniform(-decades, 0.5, size=(h, w, 1))
        img[h // 2] = img[h // 2, 0]                                     # a flat row: runs
        img[1, 1:4] = 0.0                                               # zero exponent
        return img

    layouts = [("rle", radiance(22, 31, 3), "hdr"), ("flat_wide", radiance(17, 24, 2), "hdr_flat"), ("narrow", radiance(29, 6, 2), "hdr_flat"),
               ("bytes", rng.integers(0, 256, size=(20, 26, 3), dtype=np.uint8), "hdr")]
    textures = {"floor_kd.hdr": (radiance(40, 64, 1), "hdr")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.hdr")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 2, k // 2
        x0, y0 = -3.2 + col * 3.3, 0.15 + row * 1.45
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 3.0, y0, z), (x0 + 3.0, y0 + 1.25, z), (x0, y0 + 1.25, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".hdr"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".hdr"))
    return ObjScene(
        name="hdr_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


def pic_gallery() -> ObjScene:
    """Softimage PIC diffuse maps: raw, pure run-length and mixed run-length packets, RGB and RGB + a separate alpha packet,
    a scanline-long run that needs the 16-bit count."""
    rng = np.random.default_rng(20241011)
    pos, nrm, uv = [], [], []
    groups = []

    def add(name, tris, material):
        groups.append(ObjGroup(name, _faces_same_index(np.array(tris)), material))

    add("floor", _quad(pos, nrm, uv, [(-4, 0, 3), (4, 0, 3), (4, 0, -3), (-4, 0, -3)], (0, 1, 0),
                       [(0, 0), (2, 0), (2, 1.5), (0, 1.5)]), "floor")

    def colour(h, w, alpha=False):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([40 + 200 * ((xx // 5 + yy // 3) % 2), 128 + 100 * np.sin(yy * 0.6), 30 + (6 * xx) % 200], axis=2) + rng.integers(-20, 21, size=(h, w, 3)) * (xx[:, :, None] % 9 < 4)
        img = img.clip(0, 255).astype(np.uint8)
        img[h // 2] = img[h // 2, 0]
        if alpha:
            img = np.concatenate([img, (60 + (xx // 4 * 37 + yy * 5) % 190).astype(np.uint8)[:, :, None]], axis=2)
        return img

    layouts = [("mixed", colour(21, 30), "pic"), ("raw", colour(16, 19), "pic_raw"), ("pure", colour(18, 27), "pic_pure"),
               ("mixed_a", colour(24, 33, True), "pic"), ("pure_a", colour(14, 22, True), "pic_pure"), ("long", colour(9, 300), "pic")]
    textures = {"floor_kd.pic": (colour(40, 64), "pic")}
    materials = [MtlMaterial("floor", Ns=20.0, Ka=(0.6, 0.6, 0.6), Kd=(0.9, 0.9, 0.9), Ks=(0.2, 0.2, 0.2), map_Kd="floor_kd.pic")]
    for k, (name, img, enc) in enumerate(layouts):
        col, row = k % 3, k // 3
        x0, y0 = -3.3 + col * 2.3, 0.15 + row * 1.45
        z = -1.5 - 0.25 * row
        add("panel_" + name, _quad(pos, nrm, uv, [(x0, y0, z), (x0 + 2.0, y0, z), (x0 + 2.0, y0 + 1.25, z), (x0, y0 + 1.25, z)], (0, 0, 1),
                                   [(0, 0), (1, 0), (1, 1), (0, 1)]), name)
        textures[name + ".pic"] = (img, enc)
        materials.append(MtlMaterial(name, Ns=25.0, d=1.0, Ka=(0.7, 0.7, 0.7), Kd=(1.0, 1.0, 1.0), Ks=(0.15, 0.15, 0.15), map_Kd=name + ".pic"))
    return ObjScene(
        name="pic_gallery",
        positions=np.array(pos, dtype=F32), texcoords=np.array(uv, dtype=F32), normals=np.array(nrm, dtype=F32),
        groups=groups, materials=materials, textures=textures,
        camera_position=(0.0, 1.7, 4.4), camera_facing=(0.0, -0.12, -1.0), fov=62.0)


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
    "textured_gallery": lambda: textured_gallery(),           # 192 tris, 7 materials, 14 texture files (row N1)
    "pic_gallery": lambda: pic_gallery(),                     # 14 tris, 7 materials, 7 Softimage files: raw / pure / mixed run-length, alpha packets
    "hdr_gallery": lambda: hdr_gallery(),                     # 10 tris, 5 materials, 5 Radiance files: run-length, flat, narrow
    "psd_gallery": lambda: psd_gallery(),                     # 12 tris, 6 materials, 6 PSD composites: raw / PackBits, 16-bit, every alpha value
    "gif_gallery": lambda: gif_gallery(),                     # 14 tris, 7 materials, 7 GIF files: interlaced, transparent, local tables, offset images
    "tga_gallery": lambda: tga_gallery(),                     # 16 tris, 8 materials, 8 TGA files: 5-5-5 pixels, grey + alpha, colour maps
    "bmp_gallery": lambda: bmp_gallery(),                     # 24 tris, 12 materials, 12 BMP files of every flavour the reference decodes
    "png_gallery": lambda: png_gallery(),                     # 26 tris, 13 materials, 13 PNG files: interlaced, 1 / 2 / 4-bit, colour keys
    "jpeg_gallery": lambda: jpeg_gallery(),                   # 28 tris, 14 materials, 16 JPEG files: baseline and progressive, every sampling layout
}

CONFIGS: Dict[str, RenderConfig] = {
    "C1": RenderConfig("sphere_plane", 256, 256, 1),
    "C2": RenderConfig("cornell_box", 512, 512, 4),
    "C3": RenderConfig("icosphere_l6", 1920, 1080, 8),
    "C4": RenderConfig("terrain_1m", 1920, 1080, 8),
    "C5": RenderConfig("terrain_1m", 3840, 2160, 64, bounce_depth=8),
}


def make_scene(name: str) -> ObjScene:
    return SCENES[name]()
