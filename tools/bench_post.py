#!/usr/bin/env python3
"""Times the frame post-processing kernels (bloom, bitmap packing) on a 1920x1080 frame
resident in HBM and prices them against the HBM roofline.  Prints one JSON line per kernel."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sio = importlib.import_module("ray-tracer_amd.scene_io")
capi = importlib.import_module("ray-tracer_amd.capi")


def main():
    import torch
    w, h, reps = 1920, 1080, 50
    ctx = capi.Context(0)
    rng = np.random.default_rng(1)
    frame = torch.from_numpy((rng.random((h, w, 3), dtype=np.float32) * np.float32(1.5)).astype(np.float32)).cuda()
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    for k in (3, 9):
        ctx.set_features(sio.default_features(enableBloomEffect=1, bloomFilterSize=k, bloomThreshold=0.5))
        work = frame.clone()
        ms = timed(lambda: ctx.postprocess(work.data_ptr(), w, h, stream))
        # highpass: 12 B in + 12 B out; apply: 12 B frame in + 12 B thresholded in (neighbours cache-served) + 12 B out
        alg = w * h * 60.0
        print(json.dumps({"kernel": f"bloom k={k} (highpass + apply)", "ms": round(ms, 4), "algorithmic_bytes": alg,
                          "achieved_GBps": round(alg / ms / 1e6, 1), "hbm_peak_GBps": 8000.0, "frac": round(alg / ms / 1e6 / 8000.0, 4)}))
    ms = timed(lambda: ctx.pack_bmp(frame.data_ptr(), w, h, out.data_ptr(), stream))
    alg = w * h * 16.0
    print(json.dumps({"kernel": "packBmp", "ms": round(ms, 4), "algorithmic_bytes": alg, "achieved_GBps": round(alg / ms / 1e6, 1),
                      "hbm_peak_GBps": 8000.0, "frac": round(alg / ms / 1e6 / 8000.0, 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
