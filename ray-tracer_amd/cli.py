"""Command-line rendering, the `else` branch of the reference's main (src/main.cpp:532-581):
read a TOML configuration, load the scene, build the BVH, render one frame per configured
camera on the GPU and write `<scene>_<start time>_cam_<i>.bmp` into the output directory.

    python -m ray-tracer_amd.cli path/to/config.toml [--data-dir DIR] [--device N]
"""
from __future__ import annotations

import argparse
import os
import time

from . import config as cfgmod
from . import scene_io as sio

DEFAULT_DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")


def load_scene(cfg: cfgmod.Config):
    """src/main.cpp:545-556: a mesh file takes the configured lights, a preset brings its own."""
    if cfg.sceneIsFile:
        scene = sio.Scene()
        scene.lights = list(cfg.lights)  # loadSceneFromFile, src/scene.cpp:120-129
        scene.meshes = sio.load_mesh(cfg.scene, False, scene)
        return scene, os.path.splitext(os.path.basename(cfg.scene))[0]
    scene = sio.load_scene_prebuilt(cfg.scene, cfg.dataPath, env_map=bool(cfg.features["enableEnvironmentMap"]))
    return scene, cfgmod.serialize(cfg.scene)


def render_config(cfg: cfgmod.Config, device: int = 0, start_time: str | None = None):
    """Returns [(path, (H, W, 3) float32 frame, stats)] for the configured cameras."""
    import torch  # noqa: F401  (brings the HIP runtime the library links against)
    from . import capi

    print(cfgmod.describe(cfg))
    scene, scene_name = load_scene(cfg)
    ctx = capi.Context(device)
    try:
        ctx.upload_scene(scene, sah_binning=bool(cfg.features["enableBvhSahBinning"]))  # BVH bvh(scene, features)
        ctx.set_features(cfg.features)
        os.makedirs(cfg.outputDir, exist_ok=True)
        t0 = time.perf_counter()
        stamp = start_time or time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime())
        width, height = cfg.windowSize
        done = []
        for i, cam in enumerate(cfg.cameras):
            frame, stats = ctx.render(cam.trackball(width, height), width, height)
            path = os.path.join(cfg.outputDir, f"{scene_name}_{stamp}_cam_{i}.bmp")
            print(f"Image {i} saved to {path}")
            ctx.write_bmp(frame, path)
            done.append((path, frame, stats))
        ms = int((time.perf_counter() - t0) * 1e3)
        print(f"Rendering took {ms} ms, {len(cfg.cameras)} images rendered.")
        return done
    finally:
        ctx.close()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("config")
    ap.add_argument("--data-dir", default=DEFAULT_DATA_DIR, help="stands for the reference's compiled-in DATA_DIR")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    cfg = cfgmod.read_config_file(args.config, args.data_dir)
    if not cfg.cliRenderingEnabled:
        raise SystemExit("command_line_rendering is false: the interactive window is not part of this library")
    render_config(cfg, args.device)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
