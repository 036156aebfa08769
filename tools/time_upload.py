import sys, time, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bench
capi = importlib.import_module("ray-tracer_amd.capi")
scene, f, w, h, _ = bench.build_workload("dragon")
prims = scene.primitives()
t=time.time(); nodes, p, nl, nf = capi.bvh_build(prims); print("rt_bvh_build %.3f s"%(time.time()-t))
ctx = capi.Context(0)
for i in range(3):
    t=time.time(); ctx.upload_scene(scene, nodes, p); print("rt_scene_upload (with lights) %.3f s"%(time.time()-t))
