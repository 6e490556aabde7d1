"""Timing aid (not a test): BASELINE.json configs[2] -- the eleven suite images as ONE device-resident batch (bench.py's suite_batch leg on its
own), engines per image, digests.   usage: gpu_suite_batch.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import pngloss_amd as P  # noqa: E402

golden = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))
r = bench.run_suite_batch(P, torch, lambda: P.HipContext(0), golden)
print({k: v for k, v in r.items() if k != "images"})
for im in r["images"]:
    print("  ", im)
