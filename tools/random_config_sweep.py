"""tests/test_gpu_random_configs.py over a range of seeds beyond the committed ones.
  python tools/random_config_sweep.py <first> <last> [largest frame edge]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import io
import contextlib
import test_gpu_random_configs as t  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:  # larger frames
    t.W_RANGE, t.H_RANGE = (17, int(sys.argv[3])), (17, int(sys.argv[3]))
bad = []
for seed in range(first, last):
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            t.test_random_configuration_matches_oracle(seed)
    except AssertionError:
        bad.append(seed)
        print("FAILED", buf.getvalue().strip(), flush=True)
        print(traceback.format_exc().splitlines()[-3:], flush=True)
    if seed % 50 == 0:
        print("... seed", seed, "failures so far", bad, "|", buf.getvalue().strip()[:230], flush=True)
print("seeds", first, "to", last, "failed:", bad)
