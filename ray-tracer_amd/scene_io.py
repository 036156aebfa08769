"""Host-side scene ingest used by the Python tests and the benchmark.

Builds the plain arrays the C ABI takes (include/rt_amd.h): meshes loaded the
way the reference's loader does (framework/src/mesh.cpp:52-176: per-material
sub-meshes, vertex de-duplication, optional centre + unit-scale), PNG textures
as float RGB (framework/src/image.cpp:31-53), the preset scenes and their
lights (src/scene.cpp:5-118) and a deterministic stand-in for the missing
dragon mesh with exactly 87 130 triangles (tests/bvh_tests.cpp:749-752).

Everything here is numpy; no device code and no oracle code is involved.
"""
from __future__ import annotations

import math
import os
import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

# ---------------------------------------------------------------------------
# Record layouts shared with include/rt_amd.h (and, byte for byte, with the
# oracle's header).  Vertex / Primitive / Node equal the reference's own types
# (framework/include/framework/mesh.h:14-20, src/bvh_interface.h:14-54).
# ---------------------------------------------------------------------------
VERTEX = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("texCoord", "<f4", 2)])
PRIMITIVE = np.dtype([("meshID", "<u4"), ("v0", VERTEX), ("v1", VERTEX), ("v2", VERTEX)])
NODE = np.dtype([("lower", "<f4", 3), ("upper", "<f4", 3), ("data", "<u4", 2)])
MATERIAL = np.dtype([("kd", "<f4", 3), ("ks", "<f4", 3), ("shininess", "<f4"),
                     ("transparency", "<f4"), ("kdTexture", "<i4")])
SPHERE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", MATERIAL)])
LIGHT = np.dtype([("type", "<i4"), ("v", "<f4", 21)])
RAY = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("t", "<f4")])
HIT = np.dtype([("hit", "<i4"), ("t", "<f4"), ("primitive", "<i4"), ("material", "<i4"),
                ("normal", "<f4", 3), ("barycentricCoord", "<f4", 3), ("texCoord", "<f4", 2)])
CAMERA = np.dtype([("lookAt", "<f4", 3), ("rotation", "<f4", 3), ("distance", "<f4"),
                   ("fovy", "<f4"), ("aspect", "<f4")])
FEATURES = np.dtype([
    ("enableShading", "<i4"), ("enableReflections", "<i4"), ("enableShadows", "<i4"),
    ("enableNormalInterp", "<i4"), ("enableTextureMapping", "<i4"), ("enableAccelStructure", "<i4"),
    ("enableBilinearTextureFiltering", "<i4"), ("enableTransparency", "<i4"),
    ("enableJitteredSampling", "<i4"), ("shadingModel", "<i4"), ("numPixelSamples", "<u4"),
    ("numShadowSamples", "<u4"),
    ("enableBvhSahBinning", "<i4"), ("enableBloomEffect", "<i4"), ("enableDepthOfField", "<i4"),
    ("enableEnvironmentMap", "<i4"), ("enableGlossyReflection", "<i4"),
    ("enableMipmapTextureFiltering", "<i4"), ("enableMotionBlur", "<i4"), ("motionBlurSamples", "<i4"),
    ("movement", "<f4"), ("numGlossySamples", "<u4"), ("glossyExponent", "<f4"),
    ("bloomFilterSize", "<u4"), ("bloomThreshold", "<f4"), ("depthOfFieldDistance", "<f4"),
    ("depthOfFieldCircleDiameter", "<f4"), ("numDepthOfFieldSamples", "<u4"), ("maxRayDepth", "<i4"),
])
assert VERTEX.itemsize == 32 and PRIMITIVE.itemsize == 100 and NODE.itemsize == 32
assert MATERIAL.itemsize == 36 and LIGHT.itemsize == 88 and RAY.itemsize == 28 and HIT.itemsize == 48

LEAF_BIT = 1 << 31
LIGHT_POINT, LIGHT_SEGMENT, LIGHT_PARALLELOGRAM = 0, 1, 2
SHADING_LAMBERTIAN, SHADING_PHONG, SHADING_BLINN_PHONG, SHADING_LINEAR_GRADIENT = 0, 1, 2, 3


def default_features(**kw) -> np.ndarray:
    """`Features{}` with the reference's defaults (src/common.h:62-106)."""
    f = np.zeros((), dtype=FEATURES)
    f["numPixelSamples"] = 1
    f["numShadowSamples"] = 4
    f["numGlossySamples"] = 1
    f["glossyExponent"] = 1.0
    f["bloomFilterSize"] = 3
    f["bloomThreshold"] = 0.5
    f["depthOfFieldDistance"] = 1.0
    f["depthOfFieldCircleDiameter"] = 0.1
    f["numDepthOfFieldSamples"] = 5
    f["maxRayDepth"] = 6  # src/recursive.cpp:43
    for k, v in kw.items():
        f[k] = v
    return f


def default_camera(width: int, height: int, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0),
                   rotation_deg=(20.0, 20.0, 0.0)) -> np.ndarray:
    """`CameraConfig{}` (src/config.h:18-23) turned into Trackball state the way
    the CLI does (src/main.cpp:569-571): glm::radians on fov and rotation."""
    k = np.float32(0.01745329251994329576923690768489)
    c = np.zeros((), dtype=CAMERA)
    c["lookAt"] = look_at
    c["rotation"] = np.asarray(rotation_deg, dtype=np.float32) * k
    c["distance"] = distance
    c["fovy"] = np.float32(fov_deg) * k
    c["aspect"] = np.float32(width) / np.float32(height)  # framework/src/window.cpp:413-418
    return c


def point_light(position, color) -> np.ndarray:
    l = np.zeros((), dtype=LIGHT)
    l["type"] = LIGHT_POINT
    l["v"][0:3] = position
    l["v"][3:6] = color
    return l


def segment_light(endpoint0, endpoint1, color0, color1) -> np.ndarray:
    l = np.zeros((), dtype=LIGHT)
    l["type"] = LIGHT_SEGMENT
    l["v"][0:3], l["v"][3:6], l["v"][6:9], l["v"][9:12] = endpoint0, endpoint1, color0, color1
    return l


def parallelogram_light(v0, edge01, edge02, color0, color1, color2, color3) -> np.ndarray:
    l = np.zeros((), dtype=LIGHT)
    l["type"] = LIGHT_PARALLELOGRAM
    for i, a in enumerate((v0, edge01, edge02, color0, color1, color2, color3)):
        l["v"][3 * i:3 * i + 3] = a
    return l


# ---------------------------------------------------------------------------
# PNG -> float RGB, as stbi_load(..., STBI_rgb) followed by /255.0f
# ---------------------------------------------------------------------------
def load_png_rgb(path: str) -> np.ndarray:
    """Decode a non-interlaced PNG to an (H, W, 3) float32 array in [0, 1]."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, palette, hdr = 8, [], None, None
    while pos < len(data):
        (length,), ctype = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace:
        raise ValueError(f"{path}: interlaced PNG is not supported")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits_pp = channels * depth
    stride = (w * bits_pp + 7) // 8
    bpp = max(1, bits_pp // 8)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    rows = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = int(raw[y * (stride + 1)])
        line = raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for c in range(bpp):  # per-channel running sum == Sub filter
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        else:
            cur = line.copy()
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        rows[y] = cur
        prev = cur
    if depth == 8:
        px = rows.reshape(h, w, channels)
    elif depth == 16:
        px = rows.reshape(h, w, channels, 2)[..., 0]  # stb keeps the high byte
    else:
        bits = np.unpackbits(rows, axis=1)[:, :w * depth].reshape(h, w, depth)
        vals = np.zeros((h, w), dtype=np.uint8)
        for b in range(depth):
            vals = (vals << 1) | bits[..., b]
        if ctype == 0:
            vals = (vals.astype(np.int32) * (255 // ((1 << depth) - 1))).astype(np.uint8)
        px = vals.reshape(h, w, 1)
    if ctype == 3:
        rgb = palette[px[..., 0]]
    elif ctype in (0, 4):
        rgb = np.repeat(px[..., :1], 3, axis=2)
    else:
        rgb = px[..., :3]
    return (rgb.astype(np.float32) / np.float32(255.0)).astype(np.float32)


# ---------------------------------------------------------------------------
# OBJ / MTL
# ---------------------------------------------------------------------------
@dataclass
class Mesh:
    vertices: np.ndarray  # VERTEX[]
    triangles: np.ndarray  # (n, 3) uint32
    material: np.ndarray  # MATERIAL scalar; kdTexture indexes Scene.images


@dataclass
class Scene:
    meshes: list = field(default_factory=list)
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=SPHERE))
    lights: list = field(default_factory=list)
    images: list = field(default_factory=list)  # (H, W, 3) float32 arrays
    env_faces: list = field(default_factory=lambda: [-1] * 6)

    def primitives(self) -> np.ndarray:
        """Scene-order primitive list, as gathered by BVH::BVH (src/bvh.cpp:66-77)."""
        parts = []
        for mesh_id, m in enumerate(self.meshes):
            p = np.zeros(len(m.triangles), dtype=PRIMITIVE)
            p["meshID"] = mesh_id
            p["v0"] = m.vertices[m.triangles[:, 0]]
            p["v1"] = m.vertices[m.triangles[:, 1]]
            p["v2"] = m.vertices[m.triangles[:, 2]]
            parts.append(p)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=PRIMITIVE)

    def materials(self) -> np.ndarray:
        out = np.zeros(len(self.meshes), dtype=MATERIAL)
        for i, m in enumerate(self.meshes):
            out[i] = m.material
        return out

    def lights_array(self) -> np.ndarray:
        out = np.zeros(len(self.lights), dtype=LIGHT)
        for i, l in enumerate(self.lights):
            out[i] = l
        return out

    def add_image(self, rgb: np.ndarray) -> int:
        self.images.append(np.ascontiguousarray(rgb, dtype=np.float32))
        return len(self.images) - 1


def make_material(kd=(1.0, 1.0, 1.0), ks=(0.0, 0.0, 0.0), shininess=1.0, transparency=1.0, kd_texture=-1):
    m = np.zeros((), dtype=MATERIAL)
    m["kd"], m["ks"], m["shininess"], m["transparency"], m["kdTexture"] = kd, ks, shininess, transparency, kd_texture
    return m


def _parse_mtl(path: str) -> tuple[list, dict]:
    mats, names, cur = [], {}, None
    if not os.path.exists(path):
        return mats, names
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            tok = line.split("#")[0].split()
            if not tok:
                continue
            key = tok[0]
            if key == "newmtl":
                cur = {"kd": (0.0, 0.0, 0.0), "ks": (0.0, 0.0, 0.0), "shininess": 1.0, "dissolve": 1.0,
                       "map_kd": "", "has_d": False}
                names[" ".join(tok[1:])] = len(mats)
                mats.append(cur)
            elif cur is None:
                continue
            elif key == "Kd":
                cur["kd"] = tuple(float(x) for x in tok[1:4])
            elif key == "Ks":
                cur["ks"] = tuple(float(x) for x in tok[1:4])
            elif key == "Ns":
                cur["shininess"] = float(tok[1])
            elif key == "d":
                cur["dissolve"], cur["has_d"] = float(tok[1]), True
            elif key == "Tr" and not cur["has_d"]:
                cur["dissolve"] = 1.0 - float(tok[1])
            elif key == "map_Kd":
                cur["map_kd"] = tok[-1]
    return mats, names


def load_mesh(path: str, normalize: bool = False, scene: Scene | None = None) -> list:
    """`loadMesh` (framework/src/mesh.cpp:52-149).  Textures named by map_Kd are
    decoded into `scene.images` when a scene is given."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"File {path} does not exist.")  # mesh.cpp:54-57 throws
    base = os.path.dirname(path)
    pos, nrm, uvs = [], [], []
    shapes, cur_faces, cur_mat = [], [], -1
    mats, mat_names = [], {}

    def flush():
        nonlocal cur_faces
        if cur_faces:
            shapes.append(cur_faces)
        cur_faces = []

    with open(path, "r", errors="replace") as fh:
        for line in fh:
            tok = line.split("#")[0].split()
            if not tok:
                continue
            key = tok[0]
            if key == "v":
                pos.append([np.float32(x) for x in tok[1:4]])
            elif key == "vn":
                nrm.append([np.float32(x) for x in tok[1:4]])
            elif key == "vt":
                uvs.append([np.float32(tok[1]), np.float32(tok[2]) if len(tok) > 2 else np.float32(0)])
            elif key == "mtllib":
                mats, mat_names = _parse_mtl(os.path.join(base, tok[1]))
            elif key == "usemtl":
                cur_mat = mat_names.get(" ".join(tok[1:]), -1)
            elif key in ("o", "g"):
                flush()
            elif key == "f":
                idx = []
                for part in tok[1:]:
                    f = (part.split("/") + ["", ""])[:3]

                    def fix(s, n):
                        if s == "":
                            return -1
                        i = int(s)
                        return i - 1 if i > 0 else n + i
                    idx.append((fix(f[0], len(pos)), fix(f[2], len(nrm)), fix(f[1], len(uvs))))
                if len(idx) == 3:
                    cur_faces.append((idx[0], idx[1], idx[2], cur_mat))
                elif len(idx) == 4:
                    # tinyobjloader splits a quad along its shorter diagonal
                    p = [np.asarray(pos[i[0]], dtype=np.float32) for i in idx]
                    e02, e13 = p[2] - p[0], p[3] - p[1]
                    s02 = np.float32(e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2])
                    s13 = np.float32(e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2])
                    if s02 < s13:
                        cur_faces.append((idx[0], idx[1], idx[2], cur_mat))
                        cur_faces.append((idx[0], idx[2], idx[3], cur_mat))
                    else:
                        cur_faces.append((idx[0], idx[1], idx[3], cur_mat))
                        cur_faces.append((idx[1], idx[2], idx[3], cur_mat))
                else:
                    for k in range(1, len(idx) - 1):  # fan; the bundled scenes hold only 3- and 4-gons
                        cur_faces.append((idx[0], idx[k], idx[k + 1], cur_mat))
    flush()

    pos_a = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    nrm_a = np.asarray(nrm, dtype=np.float32).reshape(-1, 3)
    uv_a = np.asarray(uvs, dtype=np.float32).reshape(-1, 2)
    tex_cache: dict = {}
    out = []
    for faces in shapes:
        n = len(faces)
        start, prev = 0, faces[0][3]
        end = 0
        while end < n:
            if end == n - 1:
                end += 1
            elif faces[end][3] == prev:
                end += 1
                continue
            else:
                prev = faces[end][3]
            verts, cache, tris = [], {}, []
            for (a, b, c, _m) in faces[start:end]:
                p0, p1, p2 = pos_a[a[0]], pos_a[b[0]], pos_a[c[0]]
                g = np.cross(p1 - p0, p2 - p0).astype(np.float32)
                g = (g * (np.float32(1.0) / np.sqrt(np.float32(g[0] * g[0] + g[1] * g[1] + g[2] * g[2])))).astype(np.float32)
                tri = []
                for (vi, ni, ti) in (a, b, c):
                    p = pos_a[vi]
                    nn = nrm_a[ni] if (ni != -1 and len(nrm_a)) else g
                    tt = uv_a[ti] if (ti != -1 and len(uv_a)) else np.zeros(2, dtype=np.float32)
                    keyv = (p.tobytes(), np.asarray(nn, dtype=np.float32).tobytes(), tt.tobytes())
                    j = cache.get(keyv)
                    if j is None:
                        j = len(verts)
                        cache[keyv] = j
                        verts.append((p, nn, tt))
                    tri.append(j)
                tris.append(tri)
            va = np.zeros(len(verts), dtype=VERTEX)
            for j, (p, nn, tt) in enumerate(verts):
                va[j]["position"], va[j]["normal"], va[j]["texCoord"] = p, nn, tt
            mid = faces[start][3]
            if mid == -1:
                mat = make_material()
            else:
                m = mats[mid]
                tex = -1
                if m["map_kd"] and scene is not None:
                    tp = os.path.join(base, m["map_kd"])
                    if tp not in tex_cache:
                        tex_cache[tp] = scene.add_image(load_png_rgb(tp))
                    tex = tex_cache[tp]
                mat = make_material(m["kd"], m["ks"], m["shininess"], m["dissolve"], tex)
            out.append(Mesh(va, np.asarray(tris, dtype=np.uint32).reshape(-1, 3), mat))
            start = end
            end = start
    if normalize:
        center_and_scale_to_unit(out)
    return out


def center_and_scale_to_unit(meshes: list) -> None:
    """`centerAndScaleToUnitMesh` (framework/src/mesh.cpp:151-176), float32 throughout."""
    allp = np.concatenate([m.vertices["position"] for m in meshes]).astype(np.float32)
    acc = np.add.accumulate(allp, axis=0, dtype=np.float32)[-1]  # sequential float sum
    center = (acc / np.float32(len(allp))).astype(np.float32)
    d = (allp - center).astype(np.float32)
    t = (d * d).astype(np.float32)
    len2 = ((t[:, 0] + t[:, 1]).astype(np.float32) + t[:, 2]).astype(np.float32)
    max_d = np.float32(np.sqrt(len2).astype(np.float32).max())
    for m in meshes:
        m.vertices["position"] = ((m.vertices["position"] - center).astype(np.float32) / max_d).astype(np.float32)


# ---------------------------------------------------------------------------
# Preset scenes (src/scene.cpp:5-118)
# ---------------------------------------------------------------------------
SCENE_TYPES = ["SingleTriangle", "Cube", "CubeTextured", "CornellBox", "CornellBoxTransparency",
               "CornellBoxParallelogramLight", "Monkey", "Teapot", "Dragon", "Spheres", "Custom"]


def cornell_parallelogram_light() -> np.ndarray:
    """src/scene.cpp:65-73"""
    return parallelogram_light((-0.2, 0.5, 0.0), (0.4, 0.0, 0.0), (0.0, 0.0, 0.4),
                               (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1))


def load_scene_prebuilt(scene_type: str, data_dir: str, env_map: bool = False) -> Scene:
    s = Scene()
    j = lambda name: os.path.join(data_dir, name)
    if scene_type == "SingleTriangle":
        s.meshes = load_mesh(j("triangle.obj"), False, s)
        s.meshes[0].material["kd"] = (1.0, 1.0, 1.0)
        s.lights.append(point_light((-1, 1, -1), (1, 1, 1)))
    elif scene_type == "Cube":
        s.meshes = load_mesh(j("cube.obj"), False, s)
        s.lights.append(segment_light((1.5, 0.5, -0.6), (-1, 0.5, -0.5), (0.9, 0.2, 0.1), (0.2, 1, 0.3)))
    elif scene_type == "CubeTextured":
        s.meshes = load_mesh(j("cube-textured.obj"), False, s)
        s.lights.append(point_light((-1.0, 1.5, -1.0), (1, 1, 1)))
    elif scene_type in ("CornellBox", "CornellBoxTransparency", "CornellBoxParallelogramLight"):
        s.meshes = load_mesh(j("CornellBox-Mirror-Rotated.obj"), True, s)
        if scene_type == "CornellBoxTransparency":
            s.meshes[6].material = make_material((1, 0.25, 0.25), (0, 0, 0), 1.0, 0.5)
            s.meshes[5].material = make_material((0.25, 1, 0.25), (0, 0, 0), 1.0, 0.5)
        if scene_type == "CornellBoxParallelogramLight":
            s.lights.append(cornell_parallelogram_light())
        else:
            s.lights.append(point_light((0, 0.58, 0), (1, 1, 1)))
    elif scene_type == "Monkey":
        s.meshes = load_mesh(j("monkey.obj"), True, s)
        s.lights.append(point_light((-1, 1, -1), (1, 1, 1)))
        s.lights.append(point_light((1, -1, -1), (1, 1, 1)))
    elif scene_type == "Teapot":
        s.meshes = load_mesh(j("teapot.obj"), True, s)
        s.lights.append(point_light((-1, 1, -1), (1, 1, 1)))
    elif scene_type == "Dragon":
        s.meshes = [dragon_standin(s)]
        s.lights.append(point_light((-1, 1, -1), (1, 1, 1)))
    elif scene_type == "Spheres":
        sp = np.zeros(3, dtype=SPHERE)
        for i, (c, r, kd) in enumerate((((3.0, -2.0, 10.2), 1.0, (0.8, 0.2, 0.2)), ((-2.0, 2.0, 4.0), 2.0, (0.6, 0.8, 0.2)),
                                        ((0.0, 0.0, 6.0), 0.75, (0.2, 0.2, 0.8)))):
            sp[i]["center"], sp[i]["radius"], sp[i]["material"] = c, r, make_material(kd)
        s.spheres = sp
        s.lights.append(point_light((3, 0, 3), (15, 15, 15)))
    elif scene_type == "Custom":
        sp = np.zeros(3, dtype=SPHERE)
        for i, (c, kd) in enumerate((((0.0, 0.0, 0.0), (0.8, 0.2, 0.2)), ((3.0, 0.0, 0.0), (0.6, 0.8, 0.2)),
                                     ((6.0, 0.0, 0.0), (0.2, 0.2, 0.8)))):
            sp[i]["center"], sp[i]["radius"], sp[i]["material"] = c, 0.5, make_material(kd)
        s.spheres = sp
        for x in (0.0, 3.0, 6.0):
            s.lights.append(point_light((x, 15.0, 0.0), (1, 1, 1)))
    else:
        raise ValueError(f"unknown scene type {scene_type}")
    if env_map:
        env_dir = os.path.join(data_dir, "environment-map-textures")
        if os.path.isdir(env_dir):
            for i in range(6):
                s.env_faces[i] = s.add_image(load_png_rgb(os.path.join(env_dir, f"sky-{i}.png")))
        else:
            attach_synthetic_env_map(s)
    return s


def attach_synthetic_env_map(s: Scene, size: int = 64, seed: int = 7) -> None:
    """Six procedural cube faces, for machines that do not hold the sky PNGs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / np.float32(size)
    for i in range(6):
        base = rng.random(3, dtype=np.float32)
        img = np.stack([(base[c] * 0.5 + 0.5 * np.abs(np.sin((i + 1) * (xx * (3 + c) + yy * (2 + i))))).astype(np.float32)
                        for c in range(3)], axis=2)
        s.env_faces[i] = s.add_image(img.astype(np.float32))


def synthetic_checker_texture(size: int = 128, cells: int = 8) -> np.ndarray:
    """A texture of the size of the reference's data/default.png (128x128)."""
    yy, xx = np.mgrid[0:size, 0:size]
    c = (((xx * cells) // size + (yy * cells) // size) & 1).astype(np.float32)
    return np.stack([0.25 + 0.75 * c, 0.35 + 0.55 * (1 - c), 0.5 + 0.25 * c], axis=2).astype(np.float32)


def dragon_standin(scene: Scene | None = None, n_u: int = 726, n_v: int = 60, seed: int = 2024) -> Mesh:
    """Deterministic stand-in for the missing data/dragon.obj.

    A displaced (2,3) torus-knot tube of n_u*n_v quads (87 120 triangles) plus
    a 10-triangle fan under it: exactly 87 130 triangles, the count the
    reference's tests pin for the dragon (tests/bvh_tests.cpp:749-752).  The
    material is data/default.mtl's (Kd 1, Ks 0.2, Ns 10, map_Kd) with a
    procedural 128x128 texture in place of default.png.  Centred and scaled to
    the unit sphere like `loadMesh(..., true)` (src/scene.cpp:90).
    """
    rng = np.random.default_rng(seed)
    u = (np.arange(n_u, dtype=np.float64) / n_u) * 2 * math.pi
    v = (np.arange(n_v, dtype=np.float64) / n_v) * 2 * math.pi
    p, q = 2.0, 3.0
    r = 2.0 + np.cos(q * u)
    c = np.stack([r * np.cos(p * u), r * np.sin(p * u), -np.sin(q * u)], axis=1)
    dc = np.gradient(c, axis=0)
    t = dc / np.linalg.norm(dc, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(t, n1)
    bump = 0.06 * rng.standard_normal((n_u, n_v))
    rad = 0.45 + 0.12 * np.sin(5 * u)[:, None] * np.cos(3 * v)[None, :] + bump * 0.3
    pos = c[:, None, :] + rad[..., None] * (np.cos(v)[None, :, None] * n1[:, None, :] + np.sin(v)[None, :, None] * n2[:, None, :])
    pos = pos.reshape(-1, 3)
    idx = lambda i, j: (i % n_u) * n_v + (j % n_v)
    ii, jj = np.meshgrid(np.arange(n_u), np.arange(n_v), indexing="ij")
    a, b, cc, d = idx(ii, jj), idx(ii + 1, jj), idx(ii + 1, jj + 1), idx(ii, jj + 1)
    tris = np.concatenate([np.stack([a, b, cc], axis=-1).reshape(-1, 3), np.stack([a, cc, d], axis=-1).reshape(-1, 3)])
    # 10-triangle fan below the knot
    zmin = pos[:, 2].min() - 0.3
    base_c = len(pos)
    ring = [[3.2 * math.cos(2 * math.pi * k / 10), 3.2 * math.sin(2 * math.pi * k / 10), zmin] for k in range(10)]
    pos = np.concatenate([pos, np.array([[0.0, 0.0, zmin]]), np.array(ring)])
    fan = np.array([[base_c, base_c + 1 + k, base_c + 1 + (k + 1) % 10] for k in range(10)])
    tris = np.concatenate([tris, fan]).astype(np.uint32)
    assert len(tris) == 87130 or (n_u, n_v) != (726, 60)
    # smooth vertex normals
    pf = pos.astype(np.float64)
    fn = np.cross(pf[tris[:, 1]] - pf[tris[:, 0]], pf[tris[:, 2]] - pf[tris[:, 0]])
    vn = np.zeros_like(pf)
    for k in range(3):
        np.add.at(vn, tris[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)
    uv = np.zeros((len(pos), 2))
    uv[:n_u * n_v, 0] = np.repeat(np.arange(n_u) / n_u, n_v)
    uv[:n_u * n_v, 1] = np.tile(np.arange(n_v) / n_v, n_u)
    uv[n_u * n_v:, 0] = 0.5 + pos[n_u * n_v:, 0] / 8.0
    uv[n_u * n_v:, 1] = 0.5 + pos[n_u * n_v:, 1] / 8.0
    verts = np.zeros(len(pos), dtype=VERTEX)
    verts["position"], verts["normal"], verts["texCoord"] = pos.astype(np.float32), vn.astype(np.float32), uv.astype(np.float32)
    tex = scene.add_image(synthetic_checker_texture()) if scene is not None else -1
    mesh = Mesh(verts, tris, make_material((1, 1, 1), (0.2, 0.2, 0.2), 10.0, 1.0, tex))
    center_and_scale_to_unit([mesh])
    return mesh
