"""bench.py's N > 1 branch on one GPU: torch.distributed is replaced by stand-ins that do what
the collectives do for rank 0 of a world in which the other ranks never answer (their shards stay
black), so the control flow, buffer shapes and stream use of the sharded path are exercised where
only one GPU is at hand.  The printed value counts rank 0's rays only.

    python tools/mock_world.py <world> [bench.py arguments]
    python tools/mock_world.py <world> --all-shards [bench.py arguments]

With --all-shards the stand-in for the gather renders the other ranks' shards itself (same GPU, a context of
its own), so the frame rank 0 puts together is the whole frame and bench.py's comparison with the one-rank
frame (`gathered_frame_equals_one_rank_frame`) means something; timings then mean nothing.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
world = int(sys.argv[1])
all_shards = "--all-shards" in sys.argv
if all_shards:
    sys.argv.remove("--all-shards")
os.environ.update(WORLD_SIZE=str(world), RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


_others = {}  # --all-shards: what stands in for the other ranks (context, scene settings), made on first use


def gather(tensor, gather_list=None, dst=0):
    gather_list[0].copy_(tensor)  # rank 0's own shard; the others' rows keep what they held
    if all_shards:
        import bench
        if not _others:
            args = _others["args"] = sys.argv
            name = args[args.index("--workload") + 1] if "--workload" in args else "dragon"
            scene, features, w, h, _ = bench.build_workload(name)
            if "--width" in args and "--height" in args:
                w, h = int(args[args.index("--width") + 1]), int(args[args.index("--height") + 1])
            c = bench.capi.Context(0)
            c.upload_scene(scene)
            c.set_features(features)
            _others.update(ctx=c, w=w, h=h, cam=bench.sio.default_camera(w, h))
        o = _others
        stream = torch.cuda.current_stream().cuda_stream
        for r in range(1, world):
            o["ctx"].render_device(o["cam"], o["w"], o["h"], gather_list[r].data_ptr(), gather_list[r].numel(), r, world, stream)
        o["ctx"].sync(stream)


dist.init_process_group = lambda *a, **k: None
dist.barrier = lambda *a, **k: None
dist.destroy_process_group = lambda *a, **k: None
dist.all_reduce = lambda t, op=None: None  # MAX and SUM over one answering rank: unchanged
dist.gather = gather
dist.get_world_size = lambda *a, **k: world
torch.cuda.device_count = lambda: world  # bench.py checks that the node has a GPU per rank; rank 0 only ever uses device 0
sys.argv = ["bench.py", "--gpus", str(world)] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
