"""Is the reference-epsilon photon camera pass a function of its inputs?  The same frame six times, and what the fp32 selection handed
over each time (Renderer.photon_selections)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
n, size, spp = 20000, 64, 16
scene, cam, cfg = scenes.CONFIGS[name]()
scene.set_option("epsilon_policy", 1)
if len(sys.argv) > 2:
    scene.set_option("photon_skip", int(sys.argv[2]))   # 1: no volume estimate; 4096: no visibility rays
r = Renderer(scene, cam).width(size).height(size).watts(14.65 * n).gather_size(20).gather_size_volume(3).seed(7)
r.photon_map_build(n, 1)
frames, sels = [], []
for i in range(6):
    r._sample_offset = 0
    frames.append(r.seed(0).photon_sample_array(spp))
    sels.append(r.photon_selections())
for i in range(1, 6):
    print(f"run {i} against run 0: the frame differs in {(frames[0] != frames[i]).any(axis=1).sum()} pixels, the selections in {(sels[0] != sels[i]).sum()} entries")
