"""ctypes binding of the product library (ray-tracer_amd/librt_amd.so, C ABI in
include/rt_amd.h).  Fails loudly when the library is missing: there is no CPU
fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scene_io as sio

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")  # RT_AMD_LIB: A/B builds

# every symbol include/rt_amd.h declares
EXPORTS = [
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_bvh_build", "rt_scene_upload", "rt_lights_set",
    "rt_features_set", "rt_render", "rt_render_device", "rt_shard_floats", "rt_deinterleave", "rt_sync",
    "rt_intersect_batch", "rt_eval_batch", "rt_postprocess_device", "rt_pack_bmp_device", "rt_bmp_header", "rt_write_bmp",
    "rt_render_device_noalloc", "rt_ctx_set_option", "rt_bvh_build_device",
]

(EVAL_BARYCENTRIC, EVAL_INTERP_NORMAL, EVAL_INTERP_TEXCOORD, EVAL_TEX_NEAREST, EVAL_TEX_BILINEAR,
 EVAL_SEGMENT_LIGHT, EVAL_PARALLELOGRAM, EVAL_SAMPLER, EVAL_CAMERA_RAY, EVAL_SHADING) = range(10)
_EVAL_IN = [12, 12, 9, 3, 3, 2, 3, 2, 2, 20]
_EVAL_OUT = [3, 3, 2, 3, 3, 6, 6, 0, 6, 3]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt_amd error {code}: {message}")
        self.code = code


class _Image(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("inner_nodes", C.c_uint64), ("tri_tests", C.c_uint64),
                ("kernel_ms", C.c_float), ("total_ms", C.c_float), ("launches", C.c_uint32), ("batches", C.c_uint32),
                ("shadow_rays", C.c_uint64), ("shadow_inner_nodes", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("path_ms", C.c_float), ("shadow_ms", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with ray-tracer_amd/csrc/Makefile "
                               "(__graft_entry__.build()); the rendering path has no CPU fallback")
        try:
            # PyTorch ships its own HIP runtime; when both live in one process it has to be
            # the first one loaded, otherwise torch finds no device afterwards.
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        l.rt_last_error.restype = C.c_char_p
        l.rt_last_error.argtypes = [C.c_void_p]
        l.rt_shard_floats.restype = C.c_size_t
        l.rt_bmp_header.restype = C.c_size_t
        l.rt_ctx_destroy.argtypes = [C.c_void_p]
        l.rt_ctx_destroy.restype = None
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def bvh_build(prims: np.ndarray, sah_binning: bool = False):
    """BVH::BVH(scene, features): returns (nodes, primitives, num_levels, num_leaves)."""
    prims = np.ascontiguousarray(prims, dtype=sio.PRIMITIVE)
    n = len(prims)
    nodes = np.zeros(max(2, 2 * n), dtype=sio.NODE)
    out = np.zeros(n, dtype=sio.PRIMITIVE)
    nn, nl, nf = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib().rt_bvh_build(_p(prims), C.c_uint32(n), int(sah_binning), _p(nodes), _p(out),
                            C.byref(nn), C.byref(nl), C.byref(nf))
    if rc != 0:
        raise RtError(rc, lib().rt_last_error(None).decode())
    return nodes[:nn.value].copy(), out, nl.value, nf.value


def bmp_header(width, height) -> bytes:
    """The 122 bytes stbi_write_bmp puts before the pixel array of a 4-component image."""
    n = int(lib().rt_bmp_header(int(width), int(height), None, C.c_size_t(0)))
    buf = (C.c_uint8 * n)()
    lib().rt_bmp_header(int(width), int(height), buf, C.c_size_t(n))
    return bytes(buf)


def shard_floats(width, height, shard, n_shard) -> int:
    return int(lib().rt_shard_floats(width, height, shard, n_shard))


OPTIONS = {"tree": 1, "per_pixel": 2, "strict_intersect": 3, "workspace_bytes": 4, "timing": 5, "tree_build": 6, "reference_boxes": 7}  # enum rt_option


class Context:
    """One rt_ctx on one HIP device; keyword arguments are rt_ctx_set_option settings (OPTIONS)."""

    def __init__(self, device: int = 0, **options):
        self.handle = C.c_void_p()
        rc = lib().rt_ctx_create(int(device), C.byref(self.handle))
        if rc != 0:
            raise RtError(rc, lib().rt_last_error(None).decode())
        self.nodes = self.prims = None
        self.num_levels = self.num_leaves = 0
        for name, value in options.items():
            self.set_option(name, value)

    def set_option(self, name: str, value: int):
        self._check(lib().rt_ctx_set_option(self.handle, int(OPTIONS[name]), C.c_int64(int(value))))

    def close(self):
        if getattr(self, "handle", None):
            lib().rt_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, lib().rt_last_error(self.handle).decode())

    def bvh_build_device(self, prims: np.ndarray, sah_binning: bool = False):
        """BVH::BVH(scene, features) built on this context's GPU: (nodes, primitives, num_levels, num_leaves,
        primitives the host had to sort because of equal centroid coordinates)."""
        prims = np.ascontiguousarray(prims, dtype=sio.PRIMITIVE)
        n = len(prims)
        nodes = np.zeros(max(2, 2 * n), dtype=sio.NODE)
        out = np.zeros(n, dtype=sio.PRIMITIVE)
        nn, nl, nf, hs = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(lib().rt_bvh_build_device(self.handle, _p(prims), C.c_uint32(n), int(sah_binning), _p(nodes), _p(out),
                                              C.byref(nn), C.byref(nl), C.byref(nf), C.byref(hs)))
        return nodes[:nn.value].copy(), out, nl.value, nf.value, hs.value

    # -- scene ----------------------------------------------------------
    def upload_scene(self, scene, nodes=None, prims=None, sah_binning=False):
        """Build the BVH over `scene` (unless nodes/prims are given; `sah_binning` is
        Features::extra.enableBvhSahBinning), upload it with materials / textures /
        spheres, and set the scene's lights."""
        if nodes is None:
            nodes, prims, self.num_levels, self.num_leaves = bvh_build(scene.primitives(), sah_binning)
        self.nodes = np.ascontiguousarray(nodes, dtype=sio.NODE)
        self.prims = np.ascontiguousarray(prims, dtype=sio.PRIMITIVE)
        mats = scene.materials()
        keep = [np.ascontiguousarray(im, dtype=np.float32) for im in scene.images]
        imgs = (_Image * max(1, len(keep)))()
        for i, im in enumerate(keep):
            imgs[i] = _Image(im.ctypes.data, im.shape[1], im.shape[0])
        env = (C.c_int32 * 6)(*scene.env_faces)
        spheres = np.ascontiguousarray(scene.spheres, dtype=sio.SPHERE)
        gen = C.c_uint64()
        self._check(lib().rt_scene_upload(
            self.handle, _p(self.nodes), C.c_uint32(len(self.nodes)), _p(self.prims), C.c_uint32(len(self.prims)),
            _p(mats), C.c_uint32(len(mats)), imgs, C.c_uint32(len(keep)), env, _p(spheres),
            C.c_uint32(len(spheres)), C.byref(gen)))
        self.set_lights(scene.lights_array())
        return gen.value

    def set_lights(self, lights):
        lights = np.ascontiguousarray(lights, dtype=sio.LIGHT)
        self._check(lib().rt_lights_set(self.handle, _p(lights), C.c_uint32(len(lights))))

    def set_features(self, features):
        f = np.ascontiguousarray(features, dtype=sio.FEATURES)
        self._check(lib().rt_features_set(self.handle, _p(f)))

    # -- rendering ------------------------------------------------------
    def render(self, camera, width, height):
        """renderImage: returns ((H, W, 3) float32 in Screen order, stats dict)."""
        out = np.zeros((height, width, 3), dtype=np.float32)
        cam = np.ascontiguousarray(camera, dtype=sio.CAMERA)
        st = Stats()
        self._check(lib().rt_render(self.handle, _p(cam), int(width), int(height), _p(out), C.byref(st)))
        return out, st.as_dict()

    def render_device(self, camera, width, height, d_out_ptr: int, d_out_floats: int, shard=0, n_shard=1,
                      stream: int = 0, collect_stats=False):
        cam = np.ascontiguousarray(camera, dtype=sio.CAMERA)
        st = Stats()
        self._check(lib().rt_render_device(self.handle, _p(cam), int(width), int(height), int(shard), int(n_shard),
                                           C.c_void_p(d_out_ptr), C.c_size_t(d_out_floats), C.c_void_p(stream),
                                           int(bool(collect_stats)), C.byref(st)))
        return st.as_dict() if collect_stats else None

    def render_device_noalloc(self, camera, width, height, d_out_ptr: int, d_out_floats: int, shard=0, n_shard=1, stream: int = 0):
        """rt_render_device without any allocation: working buffers come from the library's static scratch."""
        cam = np.ascontiguousarray(camera, dtype=sio.CAMERA)
        self._check(lib().rt_render_device_noalloc(self.handle, _p(cam), int(width), int(height), int(shard), int(n_shard),
                                                   C.c_void_p(d_out_ptr), C.c_size_t(d_out_floats), C.c_void_p(stream)))

    def deinterleave(self, d_packed_ptr: int, width, height, n_shard, d_frame_ptr: int, stream: int = 0):
        self._check(lib().rt_deinterleave(self.handle, C.c_void_p(d_packed_ptr), int(width), int(height), int(n_shard),
                                          C.c_void_p(d_frame_ptr), C.c_void_p(stream)))

    def postprocess(self, d_frame_ptr: int, width, height, stream: int = 0):
        """postprocessImageWithBloom on a device frame, in place (no-op unless enableBloomEffect)."""
        self._check(lib().rt_postprocess_device(self.handle, C.c_void_p(d_frame_ptr), int(width), int(height), C.c_void_p(stream)))

    def pack_bmp(self, d_frame_ptr: int, width, height, d_bgra_ptr: int, stream: int = 0):
        """Screen::writeBitmapToFile's 8-bit conversion into the bitmap file's pixel array (device memory)."""
        self._check(lib().rt_pack_bmp_device(self.handle, C.c_void_p(d_frame_ptr), int(width), int(height),
                                              C.c_void_p(d_bgra_ptr), C.c_void_p(stream)))

    def write_bmp(self, image: np.ndarray, path: str):
        """Screen::writeBitmapToFile for an (H, W, 3) float32 frame in host memory."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        self._check(lib().rt_write_bmp(self.handle, _p(img), int(img.shape[1]), int(img.shape[0]), str(path).encode()))

    def sync(self, stream: int = 0):
        st = Stats()
        self._check(lib().rt_sync(self.handle, C.c_void_p(stream), C.byref(st)))
        return st.as_dict()

    # -- BVHInterface::intersect -----------------------------------------
    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=sio.RAY).copy()
        hits = np.zeros(len(rays), dtype=sio.HIT)
        self._check(lib().rt_intersect_batch(self.handle, _p(rays), C.c_uint32(len(rays)), _p(hits)))
        return rays, hits

    # -- unit-tested pure functions ---------------------------------------
    def eval(self, kind: int, records, aux=None, out_count=None):
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, _EVAL_IN[kind])
        n = len(rec)
        width = _EVAL_OUT[kind] if out_count is None else out_count
        out = np.zeros((n, width), dtype=np.float32)
        auxp = _p(np.ascontiguousarray(aux)) if aux is not None else None
        self._check(lib().rt_eval_batch(self.handle, int(kind), _p(rec), C.c_uint32(n), _p(out), auxp))
        return out

    def sampler_next(self, seed: int, n: int) -> np.ndarray:
        rec = np.array([seed, n], dtype=np.uint32).view(np.float32)
        return self.eval(EVAL_SAMPLER, rec, out_count=n)[0]
