"""Kernel time against resident blocks per CU (latency-bound vs throughput-bound check).
Usage: python tools/occupancy_sweep.py [workload] [spp]"""
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
for bpc in ([int(v) for v in sys.argv[3:]] or [1, 2, 3, 4]):
    rpt_amd.set_option("blocks_per_cu", bpc)
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(8)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        r.sample_array(spp)
        ms.append(r.timing()[0])
    print(f"{name} blocks_per_cu={bpc}: kernel {min(ms):8.3f} ms  grid {r.timing()[2]}", flush=True)
