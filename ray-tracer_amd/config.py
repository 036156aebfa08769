"""The reference's TOML render configuration (src/config.h:18-43, src/config.cpp:199-390).

`read_config_file` reads the same keys with the same defaults, including the places
where the reference's reader leaves a field at its struct default, so a configuration
file written for the reference renders the same frame here:

* feature switches default to off, `num_pixel_samples` to 1, `num_shadow_samples` to
  16 (the reader's fallback, not the struct's 4), `shading_model` to 0;
* `enable_jittered_sampling` is read from `[features]` only when
  `[features.extra]` holds the key `enable_multiple_rays_per_pixel` (src/config.cpp:300-304);
* `scene` is a SceneType number, one of its names in any of the three spellings
  (src/config.cpp:425-455), or a mesh file below `data_path`;
* cameras default to fov 50, distance 3, look_at 0, rotation (20, 20, 0) degrees;
* light types `point`, `segment`, `parallelogram`; others are skipped with a message.

The reader has no keys for transparency, depth-of-field parameters, glossy samples, the
bloom parameters or the ray-depth cap (the GUI sets them, src/main.cpp:179-270).  A table
`[features.fields]` may set any Features field by its struct name (for example
`enableTransparency = true`, `maxRayDepth = 4`); the reference ignores that table.
"""
from __future__ import annotations

import dataclasses
import os
import sys

import numpy as np

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover - depends on the interpreter
    import tomli as _toml

from . import scene_io as sio

# SceneType, src/common.h (enum order) and its serialised names, src/config.cpp:392-423
SCENE_TYPES = ["SingleTriangle", "Cube", "CubeTextured", "CornellBox", "CornellBoxTransparency",
               "CornellBoxParallelogramLight", "Monkey", "Teapot", "Dragon", "Spheres", "Custom"]
_SERIALIZED = ["single_triangle", "cube", "cube_textured", "cornell_box", "cornell_box_transparency",
               "cornell_box_parallelogram_light", "monkey", "teapot", "dragon", "spheres", "custom"]


def serialize(scene_type: str) -> str:
    """src/config.cpp:392-423"""
    return _SERIALIZED[SCENE_TYPES.index(scene_type)] if scene_type in SCENE_TYPES else "unknown"


def deserialize(text: str):
    """src/config.cpp:425-455: lower-cased, with '_', '-' or nothing between the words."""
    low = text.lower()
    for name, ser in zip(SCENE_TYPES, _SERIALIZED):
        if low in (ser, ser.replace("_", ""), ser.replace("_", "-")):
            return name
    return None


@dataclasses.dataclass
class CameraConfig:  # src/config.h:18-23
    fieldOfView: float = 50.0
    distanceFromLookAt: float = 3.0
    lookAt: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (20.0, 20.0, 0.0)

    def trackball(self, width: int, height: int) -> np.ndarray:
        """src/main.cpp:569-571"""
        return sio.default_camera(width, height, self.fieldOfView, self.distanceFromLookAt, self.lookAt, self.rotation)


@dataclasses.dataclass
class Config:  # src/config.h:25-43
    cliRenderingEnabled: bool = True
    windowSize: tuple = (800, 800)
    dataPath: str = ""
    scene: object = "SingleTriangle"  # a SceneType name, or the path of a mesh file
    sceneIsFile: bool = False
    outputDir: str = ""
    features: np.ndarray = dataclasses.field(default_factory=sio.default_features)
    cameras: list = dataclasses.field(default_factory=list)
    lights: list = dataclasses.field(default_factory=list)


def _vec(v, n, default):
    if isinstance(v, (list, tuple)) and len(v) == n and all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in v):
        return tuple(float(x) for x in v)
    return default


def _flag(table, key, default=False):
    v = table.get(key, default)
    return bool(v) if isinstance(v, bool) else default


def _integer(table, key, default):
    v = table.get(key, default)
    return int(v) if isinstance(v, int) and not isinstance(v, bool) else default


def read_config_file(path: str, default_data_dir: str) -> Config:
    """readConfigFile, src/config.cpp:199-390.  `default_data_dir` stands for the
    reference's compiled-in DATA_DIR."""
    with open(path, "rb") as fh:
        table = _toml.load(fh)
    cfg = Config()
    cfg.cliRenderingEnabled = _flag(table, "command_line_rendering", True)
    ws = table.get("window_size")
    if isinstance(ws, list) and len(ws) == 2 and all(isinstance(x, int) and not isinstance(x, bool) for x in ws):
        cfg.windowSize = (int(ws[0]), int(ws[1]))
    data_path = table.get("data_path", default_data_dir)
    if not isinstance(data_path, str) or data_path == "default":
        data_path = default_data_dir
    cfg.dataPath = data_path

    scene = table.get("scene")
    if isinstance(scene, int) and not isinstance(scene, bool):
        if not 0 <= scene < len(SCENE_TYPES):
            raise ValueError(f"scene type {scene} is not a SceneType")
        cfg.scene = SCENE_TYPES[scene]
    else:
        text = scene if isinstance(scene, str) else "none"
        name = deserialize(text)
        if name is not None:
            cfg.scene = name
        else:
            full = os.path.join(cfg.dataPath, text)
            if not os.path.exists(full):
                raise FileNotFoundError(f"Error: Scene file {full} does not exist.")
            cfg.scene, cfg.sceneIsFile = full, True

    out = table.get("output_dir", "")
    if not isinstance(out, str) or not out:
        print("Warning: No output directory specified, using current directory.")
        cfg.outputDir = os.getcwd()
    else:
        home = os.environ.get("HOME", "")
        if out.startswith("~"):
            out = home + out[1:]
        if out.startswith("$HOME"):
            out = home + out[5:]
        cfg.outputDir = os.path.abspath(out)

    ft = table.get("features", {}) if isinstance(table.get("features"), dict) else {}
    extra = ft.get("extra", {}) if isinstance(ft.get("extra"), dict) else {}
    f = cfg.features
    f["enableShading"] = _flag(ft, "enable_shading")
    f["enableReflections"] = _flag(ft, "enable_reflections")
    f["enableShadows"] = _flag(ft, "enable_shadows")
    f["enableNormalInterp"] = _flag(ft, "enable_normal_interp")
    f["enableTextureMapping"] = _flag(ft, "enable_texture_mapping")
    f["enableAccelStructure"] = _flag(ft, "enable_accel_structure")
    f["numPixelSamples"] = _integer(ft, "num_pixel_samples", 1)
    f["shadingModel"] = _integer(ft, "shading_model", 0)
    f["numShadowSamples"] = _integer(ft, "num_shadow_samples", 16)
    if "enable_bilinear_texture_filtering" in ft:
        f["enableBilinearTextureFiltering"] = _flag(ft, "enable_bilinear_texture_filtering")
    if "enable_bloom_effect" in extra:
        f["enableBloomEffect"] = _flag(extra, "enable_bloom_effect")
    f["enableEnvironmentMap"] = _flag(extra, "enable_environment_map")
    if "enable_multiple_rays_per_pixel" in extra:
        f["enableJitteredSampling"] = _flag(ft, "enable_jittered_sampling")
    for key, field in (("enable_motion_blur", "enableMotionBlur"), ("enable_depth_of_field", "enableDepthOfField"),
                       ("enable_glossy_reflection", "enableGlossyReflection"),
                       ("enable_mipmap_texture_filtering", "enableMipmapTextureFiltering")):
        if key in extra:
            f[field] = _flag(extra, key)
    fields = ft.get("fields", {}) if isinstance(ft.get("fields"), dict) else {}
    for name, value in fields.items():  # not read by the reference: Features fields by struct name
        if name not in sio.FEATURES.names:
            raise KeyError(f"[features.fields]: Features has no field {name}")
        f[name] = value

    for cam in table.get("cameras", []) if isinstance(table.get("cameras"), list) else []:
        fov = cam.get("field_of_view", 50.0)
        dist = cam.get("distance_from_look_at", 3.0)
        cfg.cameras.append(CameraConfig(float(fov) if isinstance(fov, float) else 50.0,
                                        float(dist) if isinstance(dist, float) else 3.0,
                                        _vec(cam.get("look_at"), 3, (0.0, 0.0, 0.0)),
                                        _vec(cam.get("rotation"), 3, (20.0, 20.0, 0.0))))

    lights = table.get("lights")
    if isinstance(lights, list):
        zero = (0.0, 0.0, 0.0)
        for light in lights:
            kind = light.get("type", "none")
            if kind == "point":
                cfg.lights.append(sio.point_light(_vec(light.get("position"), 3, zero), _vec(light.get("color"), 3, zero)))
            elif kind == "segment":
                e, c = light.get("endpoints", [None, None]), light.get("colors", [None, None])
                cfg.lights.append(sio.segment_light(_vec(e[0], 3, zero), _vec(e[1], 3, zero), _vec(c[0], 3, zero), _vec(c[1], 3, zero)))
            elif kind == "parallelogram":
                e, c = light.get("edges", [None, None]), light.get("colors", [None] * 4)
                cfg.lights.append(sio.parallelogram_light(_vec(light.get("corner"), 3, zero), _vec(e[0], 3, zero), _vec(e[1], 3, zero),
                                                          *[_vec(c[i], 3, zero) for i in range(4)]))
            else:
                print(f"Unknown light type: {kind} -- Skip", file=sys.stderr)
    else:
        print("WARN: No lights found in config file.", file=sys.stderr)
    return cfg


def describe(cfg: Config) -> str:
    """What the command-line renderer prints before it starts (operator<<(Config), src/config.cpp:13-196), abridged
    to one line per entry."""
    f = cfg.features
    lines = [f"command_line_rendering: {cfg.cliRenderingEnabled}", f"window_size: {cfg.windowSize[0]} x {cfg.windowSize[1]}",
             f"data_path: {cfg.dataPath}", f"scene: {cfg.scene if cfg.sceneIsFile else serialize(cfg.scene)}",
             f"output_dir: {cfg.outputDir}", "features:"]
    lines += [f"  {name}: {f[name]}" for name in sio.FEATURES.names]
    lines += [f"camera {i}: fov {c.fieldOfView} distance {c.distanceFromLookAt} look_at {c.lookAt} rotation {c.rotation}"
              for i, c in enumerate(cfg.cameras)]
    lines.append(f"lights: {len(cfg.lights)}")
    return "\n".join(lines)
