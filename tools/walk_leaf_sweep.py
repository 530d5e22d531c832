"""Kernel time of a per-mesh-tree scene against "walk_leaf_quarters" (deferred walks: the descent pauses for the leaves when
4 x lanes-at-a-leaf >= this x lanes-descending; 0 = when every lane is at a leaf).
Usage: python tools/walk_leaf_sweep.py [workload] [width] [spp] [values ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
values = [int(v) for v in sys.argv[4:]] or [0, 16, 8, 6, 4, 3, 2, 1]
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
ref = None
for v in values:
    rpt_amd.set_option("walk_leaf_quarters", v)
    r = Renderer(scene, cam).width(width).height(width).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        img = r.sample_array(spp)
        ms.append(r.timing()[0])
    if ref is None:
        ref = img
    print(f"{name} {width}x{width}x{spp} walk_leaf_quarters={v:3d}: kernel {min(ms):9.3f} ms   identical to first: {np.array_equal(img, ref)}", flush=True)
rpt_amd.set_option("walk_leaf_quarters", 6)
