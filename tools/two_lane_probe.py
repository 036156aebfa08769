#!/usr/bin/env python3
"""Probe: frames rendered alternately by two contexts on two streams (two frames in flight)
against one context on one stream.  Prints ms per frame for both."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sio = importlib.import_module("ray-tracer_amd.scene_io")
capi = importlib.import_module("ray-tracer_amd.capi")
import torch

def run(lanes, width, height, steps=20, warmup=4):
    scene, features, _, _, _ = bench.build_workload("dragon")
    cam = sio.default_camera(width, height)
    dev = torch.device("cuda", 0)
    ctxs, streams, frames = [], [], []
    for _ in range(lanes):
        c = capi.Context(0); c.upload_scene(scene); c.set_features(features)
        ctxs.append(c); streams.append(torch.cuda.Stream(dev)); frames.append(torch.empty(width * height * 3, dtype=torch.float32, device=dev))
    def step(i):
        k = i % lanes
        ctxs[k].render_device(cam, width, height, frames[k].data_ptr(), frames[k].numel(), 0, 1, streams[k].cuda_stream)
    for i in range(warmup): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    for c in ctxs: c.close()
    return dt

for (w, h) in [(1920, 1080), (960, 540), (680, 382)]:
    print(f"{w}x{h}: " + ", ".join(f"{n} in flight {run(n, w, h):.3f} ms/frame" for n in (1, 2, 3, 4)))
