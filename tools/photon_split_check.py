"""C4 camera pass: one launch against the split form (volume estimate, then surface estimate; option "photon_split").
Usage: python tools/photon_split_check.py [spp]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene, cam, cfg = scenes.CONFIGS["C4"]()
n = cfg["photons"]
rpt_amd.set_option("timing", 1)
frames = {}
for split in (0, 1):
    rpt_amd.set_option("photon_split", split)
    r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
    r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
    r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        frames[split] = r.photon_sample_array(spp)
        ms.append(r.timing()[0])
    print(f"photon_split={split}: camera pass {min(ms):9.3f} ms for {cfg['width']}x{cfg['height']}x{spp}, mean {frames[split].mean():.9f}", flush=True)
d = frames[1] - frames[0]
print(f"split vs one launch: rel RMS {np.sqrt((d ** 2).mean() / (frames[0] ** 2).mean()):.3e}, max abs {np.abs(d).max():.3e}")
