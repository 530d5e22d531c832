"""Wave-level counters of the C4 surface gather (counters build).  Usage: python tools/photon_gather_counters.py [spp]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, _lib, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene, cam, cfg = scenes.CONFIGS["C4"]()
n = cfg["photons"]
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
print(r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM))
rpt_amd.set_option("counters", 1)
r.photon_sample_array(spp)
c = r.counters()
out = (C.c_uint64 * 56)()
_lib.check(_lib.load().rpt_debug_section_counters(r.scene._handle, out))
names = ["trips with a gather", "terms without a scan", "ball-walk steps", "candidates", "overfull walks", "selection steps",
         "list updates", "second-pass candidates", "second-pass photon terms", "new anchors",
         "trips searching one by one", "lanes searching one by one"]
trips = max(int(out[0]), 1)
print(f"samples {c['samples']}, beam tests per sample {c['bvh_nodes'] / c['samples']:.1f}")
for k, nm in enumerate(names):
    print(f"  {nm:32s} {int(out[k]):12d}   per trip {int(out[k]) / trips:8.2f}")
tn = ["first pass over the rays", "beam estimate", "pixel candidate list", "second pass over the rays",
      "hit record + material (+ volume estimate, samples in lanes)", "surface gather", "pixel sum"]
tot = sum(int(out[16 + k]) for k in range(len(tn))) or 1
print("wave clock ticks per part (share of the pixels' time):")
for k, nm in enumerate(tn):
    print(f"  {nm:60s} {100.0 * int(out[16 + k]) / tot:6.2f} %")
for k, nm in ((7, "selection"), (8, "scan mask, thresholds"), (9, "terms")):
    print(f"    of the surface gather: {nm:36s} {100.0 * int(out[16 + k]) / tot:6.2f} %")
