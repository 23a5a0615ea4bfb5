#!/usr/bin/env python3
"""profiles/traffic.json quotes hardware-counter constants (bytes, missed lines, instructions per launch) that
bench.py cannot measure itself.  They are valid for the kernel sources they were taken on: this script prints
the hash of those sources (bench.KERNEL_SOURCES) and, with --update, stamps it into traffic.json — to be run
together with the counter passes (tools/prof_r03.sh), never on its own after a kernel change."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

h = bench.kernel_sources_hash()
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
print("this tree:", h, " stored:", tj.get("kernel_sources_sha256_16"))
if "--update" in sys.argv:
    tj["kernel_sources_sha256_16"] = h
    tj["kernel_sources"] = bench.KERNEL_SOURCES
    json.dump(tj, open(path, "w"), indent=1)
    print("stamped")
