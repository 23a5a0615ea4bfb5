#!/usr/bin/env python3
"""profiles/traffic.json from the counter passes tools/prof_r03.sh took (condensed by tools/summarize_prof.py into
profiles/<tag>_pmc.json and profiles/<tag>_s500_pmc.json), stamped with the hash of the kernel sources they were
taken on.  usage: tools/make_traffic.py TAG [commit]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
import valu_mix  # noqa: E402

KERNELS = {"bounce": "trace_kernel<true, false, 0, 8, 256, false, false>",
           "bounce_s500": "trace_kernel<true, false, 0, 8, 256, false, false>",
           "cornell_shadow": "trace_kernel<true, true, 2, 8, 256, false, false>",
           "transmittance": "transm_shadow_kernel"}


def mix(key):
    """Static vector-ALU mix of the kernel (tools/valu_mix.py): the mean issue cost bench.py prices an instruction at."""
    k = valu_mix.static_mix(KERNELS[key])[0]
    return {"valu_static_mix": {"kernel": k["name"], "instructions": k["n"],
                                "by_cycles": {str(c): v for c, v in sorted(k["by_cost"].items())},
                                "mean_cycles": round(k["mean_cycles"], 3),
                                "source": "tools/valu_mix.py on this tree; issue costs from profiles/r03_valu_issue.txt"}}


tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else bench.git_head()
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
N = float(1 << 24)
for key, t, note in (("bounce", tag, "S(158), 2^24 rays, leaf blocks n>=6"),
                     ("bounce_s500", tag + "_s500", "S(500), 2^24 rays, leaf blocks n>=2; scaled per ray for other batch sizes")):
    f = os.path.join(ROOT, "profiles", "%s_pmc.json" % t)
    if not os.path.exists(f):
        print("missing", f)
        continue
    c = json.load(open(f))["counters"]
    miss = c["TCC_MISS_sum"]["per_launch_mean"]
    wr = c["WRITE_SIZE"]["per_launch_mean"] * 1024
    ent = {"hbm_bytes_per_launch": miss * 128 + wr, "layout": 2, "source": "profiles/%s_pmc.json" % t, "commit": commit,
           "fetch_size_bytes": c["FETCH_SIZE"]["per_launch_mean"] * 1024, "write_size_bytes": wr,
           "tcc_miss_per_launch": miss, "tcc_miss_per_ray": miss / N, "note": note}
    if "SQ_INSTS_VALU" in c:
        ent["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]["per_launch_mean"]
        ent["valu_insts_per_pair"] = c["SQ_INSTS_VALU"]["per_launch_mean"] / N
        ent.update(mix(key))
    tj[key] = ent
    print(key, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ent.items()})
import re
for key, pairs in (("cornell_shadow", 152059200), ("transmittance", 152059200)):   # 1,013,728 hit pixels x 150 golden VPLs
    f = os.path.join(ROOT, "profiles", "%s_%s_counters.txt" % (tag.split("_")[0], key))
    if not os.path.exists(f):
        print("missing", f)
        continue
    m = re.search(r"SQ_INSTS_VALU\s+([0-9.e+]+) per launch", open(f).read())
    v = float(m.group(1))
    tj[key] = {"valu_insts_per_launch": v, "pairs": pairs, "valu_insts_per_pair": v / pairs,
               "source": "profiles/%s_%s_counters.txt (tools/prof_extras_r03.sh)" % (tag.split("_")[0], key), "commit": commit}
    tj[key].update(mix(key))
    print(key, tj[key])
tj["kernel_sources_sha256_16"] = bench.kernel_sources_hash()
tj["kernel_sources"] = bench.KERNEL_SOURCES
json.dump(tj, open(path, "w"), indent=1)
print("stamped", tj["kernel_sources_sha256_16"])
