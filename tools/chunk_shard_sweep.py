"""Render-kernel time against samples per work item ("chunk_spp") for the whole frame and for one rank's shard of an
N-GPU run: small items keep the tail of a 1/8 shard short, large items write fewer partial sums.
Usage: python tools/chunk_shard_sweep.py [workload] [spp] [shard counts...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
counts = [int(v) for v in sys.argv[3:]] or [1, 8]
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
for count in counts:
    for v in (2, 4, 8, 16, 32):
        rpt_amd.set_option("chunk_spp", v)
        r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0).shard(0, count)
        r.sample_array(8)
        ms = []
        for _ in range(4):
            r._sample_offset = 0
            r.sample_array(spp)
            ms.append(r.timing()[0])
        print(f"{name} shard 0/{count} chunk_spp={v:2d}: kernel {min(ms):8.3f} ms", flush=True)
rpt_amd.set_option("chunk_spp", 0)
