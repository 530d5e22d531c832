"""C4 camera pass with and without the sort of the strips' candidate lists (photon_skip bit 1024 leaves them in walk order) and
with the surface gather's radius guess kept from one work item to the next (bit 2048) instead of starting afresh.
Usage: python tools/photon_sort_check.py [spp]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene, cam, cfg = scenes.CONFIGS["C4"]()
rpt_amd.set_option("timing", 1)
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
r.photon_map_build(cfg["photons"], Renderer.PHOTON_POINT_BEAM)
frames = {}
for skip in (1024, 0, 2048, 1024, 0, 2048):
    rpt_amd.set_option("photon_skip", skip)
    ms = []
    for _ in range(3):
        r._sample_offset = 0
        frames[skip] = r.photon_sample_array(spp)
        ms.append(r.timing()[0])
    print(f"{ {0: 'lists sorted, guess per item   ', 1024: 'lists in walk order            ', 2048: 'guess kept across work items   '}[skip]}: camera pass {min(ms):9.3f} ms, mean {frames[skip].mean():.9f}", flush=True)
rpt_amd.set_option("photon_skip", 0)
d = frames[0] - frames[1024]
print(f"sorted vs walk order: rel RMS {np.sqrt((d ** 2).mean() / (frames[0] ** 2).mean()):.3e}")
