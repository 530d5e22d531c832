"""Kernel time against samples per pixel on one rank's shard: time = fixed + slope * spp.  The fixed part (grid start,
first trips in lock-step, the tail of the last paths) is what multi-GPU scaling loses.
Usage: python tools/fixed_cost.py [workload] [shard_count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0).shard(0, count)
xs, ys = [], []
for spp in (4, 8, 16, 32, 64, 128, 256):
    best = 1e9
    for _ in range(4):
        r._sample_offset = 0
        r.sample_array(spp)
        t = r.timing()
        best = min(best, t[0])
    xs.append(spp)
    ys.append(best)
    print(f"{name} shard 0/{count} spp={spp:4d}: render kernel {best:8.3f} ms  (resolve {t[1]:.3f} ms)", flush=True)
slope, fixed = np.polyfit(xs, ys, 1)
print(f"fit: {fixed:.3f} ms + {slope:.5f} ms/spp")
