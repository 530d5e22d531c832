"""Render one workload a few times (for profilers that sample a running kernel).
Usage: python tools/render_loop.py [workload] [frames] [width] [spp]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scene, cam, cfg = scenes.CONFIGS[name]()
width = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["width"]
spp = int(sys.argv[4]) if len(sys.argv) > 4 else cfg["spp"]
r = Renderer(scene, cam).width(width).height(width * cfg["height"] // cfg["width"]).max_bounces(cfg.get("max_bounces", 0)).seed(0)
if name == "C4":
    r.gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).watts(cfg["renderer_watts"])
    r.photon_map_build(cfg["photons"], Renderer.PHOTON_POINT_BEAM)
for _ in range(frames):
    r._sample_offset = 0
    img = r.photon_sample_array(spp) if name == "C4" else r.sample_array(spp)
print(name, width, spp, "mean", float(img.mean()))
