"""Per-sample counts of the C4 beam query: photons tested per ray (lane tests) and accepted.
Usage: python tools/photon_counters.py [spp]"""
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scene, cam, cfg = scenes.CONFIGS["C4"]()
n = cfg["photons"]
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).seed(0)
r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
print(r.photon_map_build(n, Renderer.PHOTON_POINT_BEAM))
rpt_amd.set_option("counters", 1)
r.photon_sample_array(spp)
c = r.counters()
print(c)
s = c["samples"]
print(f"lane tests per sample {c['bvh_nodes'] / s:.1f}, accepted per sample {c['bvh_tris'] / s:.2f}")
