"""One counting frame of a bench workload with the per-wave start/end log printed
(RT_AMD_DEBUG_COUNTERS): when the waves of the two traversal kernels start and end."""
import importlib
import os
import sys

os.environ["RT_AMD_DEBUG_COUNTERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before the library: it brings the HIP runtime)
import bench  # noqa: E402

capi = importlib.import_module("ray-tracer_amd.capi")
sio = importlib.import_module("ray-tracer_amd.scene_io")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene, features, w, h, desc = bench.build_workload(name)
if len(sys.argv) > 3:
    w, h = int(sys.argv[2]), int(sys.argv[3])
ctx = capi.Context(0)
ctx.upload_scene(scene)
ctx.set_features(features)
cam = sio.default_camera(w, h)
out = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
for _ in range(2):
    ctx.render_device(cam, w, h, out.data_ptr(), out.numel(), 0, 1, 0)
ctx.sync(0)
st = ctx.render_device(cam, w, h, out.data_ptr(), out.numel(), 0, 1, 0, collect_stats=True)
print(desc, {k: st[k] for k in ("rays", "shadow_rays", "kernel_ms", "path_ms", "shadow_ms")})
