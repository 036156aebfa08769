#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the counters tools/pmc.sh collected.

    python3 tools/pmc_summarise.py <dir>          # text summary on stdout
    python3 tools/pmc_summarise.py <dir> --json   # {"kernel": {"counter": mean per launch}}
    python3 tools/pmc_summarise.py <dir> --traffic  # {"kernel": {"fetch_bytes", "write_bytes"}} per launch
    python3 tools/pmc_summarise.py <dir> --counters <workload> <summary-file>
                                                  # what bench.py reads: profiles/pmc_counters_<workload>.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"rt::(\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def collect(root: str):
    out = {}
    for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        pas = os.path.basename(os.path.dirname(f))
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        out[pas] = {k: {c: sum(d.values()) / len(d) for c, d in cs.items()} for k, cs in acc.items()}
    return out


def sources_sha1() -> str:
    """Hash of the sources the queue-based renderer's kernels are compiled from (rt_wavefront.hip and what it
    includes): tells bench.py whether stored counters belong to the kernels it runs."""
    import hashlib
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray-tracer_amd", "csrc")
    h = hashlib.sha1()
    for f in ("rt_device.hpp", "rt_fastdiv.hpp", "rt_pow.hpp", "rt_trace.hpp", "rt_trig.hpp", "rt_wavefront.hip", "rt_wavefront.hpp"):
        h.update(f.encode())
        h.update(open(os.path.join(here, f), "rb").read())
    return h.hexdigest()


def main():
    root = sys.argv[1]
    data = collect(root)
    if "--counters" in sys.argv:
        at = sys.argv.index("--counters")
        workload, summary = sys.argv[at + 1], sys.argv[at + 2]
        merged = defaultdict(dict)
        for pas in data.values():
            for k, cs in pas.items():
                if k.startswith("__amd") or re.search(r"<true", k):  # (counting variants: first template argument)
                    continue
                merged[re.sub(r"<.*", "", k)].update(cs)
        for k, cs in merged.items():
            # FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled
            # (an upper bound for narrower requests)
            if "FETCH_SIZE" in cs:
                cs["fetch_bytes"] = cs["FETCH_SIZE"] * 2048.0
            if "WRITE_SIZE" in cs:
                cs["write_bytes"] = cs["WRITE_SIZE"] * 1024.0
        print(json.dumps({"workload": workload, "summary": summary, "sources_sha1": sources_sha1(),
                          "source": "rocprofv3 --pmc, one run per pass (tools/pmc.sh), mean per launch of the non-counting kernels; "
                                    "FETCH_SIZE doubled as gfx950 tallies 128-B requests at 64 B",
                          "kernels": merged}, indent=1, sort_keys=True))
        return
    if "--traffic" in sys.argv:
        # HBM-side bytes per launch: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies
        # 128-B requests at 64 B, so it is doubled (an upper bound for narrower requests)
        t = {}
        for k, cs in data.get("fetch", {}).items():
            if k.startswith("__amd") or re.search(r"<true", k):  # (counting variants: first template argument)
                continue
            w = data.get("write", {}).get(k, {}).get("WRITE_SIZE", 0.0)
            t[re.sub(r"<.*", "", k)] = {"fetch_bytes": cs["FETCH_SIZE"] * 2048.0, "write_bytes": w * 1024.0}
        print(json.dumps(t, indent=1, sort_keys=True))
        return
    if "--json" in sys.argv:
        merged = defaultdict(dict)
        for pas in data.values():
            for k, cs in pas.items():
                merged[k].update(cs)
        print(json.dumps(merged, indent=1, sort_keys=True))
        return
    for pas, ks in data.items():
        print(f"## pass {pas} (mean per launch)")
        for k in sorted(ks):
            if k.startswith("__amd"):
                continue
            print(f"{k}: " + ", ".join(f"{c}={v:.4g}" for c, v in sorted(ks[k].items())))
    print("## derived")
    for k in sorted(data.get("fetch", {})):
        if k.startswith("__amd"):
            continue
        line = [k]
        fs = data["fetch"][k].get("FETCH_SIZE")
        ws = data.get("write", {}).get(k, {}).get("WRITE_SIZE")
        if fs is not None:
            # FETCH_SIZE is in KiB and counts 128-B requests at 64 B on gfx950: doubled here
            line.append(f"fetch_bytes(x2 corrected)={fs * 1024 * 2:.4g}")
        if ws is not None:
            line.append(f"write_bytes={ws * 1024:.4g}")
        w = data.get("write", {}).get(k, {})
        if "TCC_HIT_sum" in w and w["TCC_HIT_sum"] + w.get("TCC_MISS_sum", 0) > 0:
            line.append(f"L2_hit={w['TCC_HIT_sum'] / (w['TCC_HIT_sum'] + w['TCC_MISS_sum']):.3f}")
        s = data.get("sq", {}).get(k, {})
        if s.get("SQ_INSTS_VALU"):
            line.append(f"lanes_per_valu={s['SQ_THREAD_CYCLES_VALU'] / s['SQ_INSTS_VALU']:.1f}")
        m = data.get("mix", {}).get(k, {})
        if m.get("GRBM_GUI_ACTIVE"):
            line.append(f"TA_busy={m['TA_BUSY_avr'] / (m['GRBM_GUI_ACTIVE'] / 8):.2f}")
        print(" ".join(line))


if __name__ == "__main__":
    main()
