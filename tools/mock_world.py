"""bench.py's N > 1 branch on one GPU: torch.distributed is replaced by stand-ins that do what
the collectives do for rank 0 of a world in which the other ranks never answer (their shards stay
black), so the control flow, buffer shapes and stream use of the sharded path are exercised where
only one GPU is at hand.  The printed value counts rank 0's rays only.

    python tools/mock_world.py <world> [bench.py arguments]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
world = int(sys.argv[1])
os.environ.update(WORLD_SIZE=str(world), RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def gather(tensor, gather_list=None, dst=0):
    gather_list[0].copy_(tensor)  # rank 0's own shard; the others' rows keep what they held


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
