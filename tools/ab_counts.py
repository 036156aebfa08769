"""Ray, node and triangle counts of the headline frame (counting launch), for A/B builds (RT_AMD_LIB)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
capi = importlib.import_module("ray-tracer_amd.capi")
sio = importlib.import_module("ray-tracer_amd.scene_io")
scene, features, w, h, desc = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "dragon")
cam = sio.default_camera(w, h)
c = capi.Context(0)
c.upload_scene(scene); c.set_features(features)
frame = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
st = c.render_device(cam, w, h, frame.data_ptr(), frame.numel(), 0, 1, 0, collect_stats=True)
print(json.dumps(st))
