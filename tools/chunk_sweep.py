"""Kernel time against samples per work item ("chunk_spp").  Usage: python tools/chunk_sweep.py [workload] [spp] [values...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
values = [int(v) for v in sys.argv[3:]] or [0, 2, 4, 8, 16, 32]
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
ref = None
for v in values:
    rpt_amd.set_option("chunk_spp", v)
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(8)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        img = r.sample_array(spp)
        ms.append(r.timing()[0])
    if ref is None:
        ref = img
    print(f"{name} chunk_spp={v:2d}: kernel {min(ms):8.3f} ms   max |diff| vs first {np.abs(img - ref).max():.3e}", flush=True)
