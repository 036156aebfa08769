import sys, numpy as np, importlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import common
from common import sio
name = sys.argv[1]
scn = common.scene(name)
p = scn.primitives()
v = np.stack([p["v0"]["position"], p["v1"]["position"], p["v2"]["position"]], axis=1)  # n,3,3
lo = v.min(axis=1); hi = v.max(axis=1)
np.concatenate([lo, hi], axis=1).astype(np.float32).tofile(f"/tmp/{name}.bin")
print(name, len(p))
