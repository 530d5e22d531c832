"""Kernel time of a per-mesh-tree scene against "defer_lanes" (parked tree walks per wave that trigger a walk)
and "defer_stop" (still-walking lanes below which the wave leaves the walk).
Usage: python tools/defer_sweep.py [workload] [width] [spp] [lanes:stop ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
values = [tuple(int(x) for x in v.split(":")) for v in sys.argv[4:]] or [(1, 1), (40, 1), (40, 8), (40, 16), (40, 24), (40, 32), (32, 16),
                                                                          (32, 24), (48, 16), (48, 24), (48, 32), (56, 32)]
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
ref = None
for v in values:
    rpt_amd.set_option("defer_lanes", v[0])
    rpt_amd.set_option("defer_stop", v[1])
    r = Renderer(scene, cam).width(width).height(width).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        img = r.sample_array(spp)
        ms.append(r.timing()[0])
    if ref is None:
        ref = img
    print(f"{name} {width}x{width}x{spp} defer_lanes:stop={v[0]:2d}:{v[1]:2d}: kernel {min(ms):9.3f} ms   identical to first: {np.array_equal(img, ref)}", flush=True)
