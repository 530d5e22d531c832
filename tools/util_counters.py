"""Lane utilisation of the two closest-hit scans per trip, from the device counters (COUNT build).
Usage: python tools/util_counters.py [workload] [spp]"""
import json
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene, cam, cfg = scenes.CONFIGS[name]()
r = Renderer(scene, cam).width(cfg["width"]).height(cfg["height"]).max_bounces(cfg["max_bounces"]).seed(0)
rpt_amd.set_option("counters", 1)
r.sample_array(spp)
c = r.counters()
c["rays_per_sample"] = c["rays"] / c["samples"]
c["vertices_per_sample"] = c["vertices"] / c["samples"]
c["lanes_per_trip_primary"] = c["vertices"] / c["wave_trips"]
c["lanes_per_trip_shadow"] = (c["rays"] - c["vertices"]) / c["wave_trips"]
print(json.dumps(c, indent=1))
print(json.dumps(r.scene_stats()))
