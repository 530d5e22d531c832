"""Lane utilisation per section of the megakernel's loop body (COUNT build, SECT(k) in kernels.hip).
Usage: python tools/trips.py [workload] [spp] [detach_shadows 0|1|2]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, _lib, scenes  # noqa: E402

NAMES = ["0 work pull", "1 regenerate camera ray", "2 vertex start (medium d, wo)", "3 after primary scan", "4 miss/env",
         "5 medium event setup", "6 surface finalize + material", "7 light sample", "8 shadow scan start",
         "9 after shadow scan (visibility, NEE shading)", "10 bounce start", "11 medium bounce", "12 surface RR",
         "13 surface sample_f + bsdf", "14 path update", "15 parked tree walks (per-mesh-tree kernels)", "16 walk finished", "17 ... with a triangle hit", "18 ... shadow query", "19", "20",
         "21 detached: shadow query queued / streamed: path parked", "22 detached: queue full, walked in place / streamed: shadow query written"]
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene, cam, cfg = scenes.CONFIGS[name]()
r = Renderer(scene, cam).width(min(cfg["width"], 1024)).height(min(cfg["height"], 1024)).max_bounces(cfg["max_bounces"]).seed(0)
if len(sys.argv) > 3:
    rpt_amd.set_option("detach_shadows", int(sys.argv[3]))
import os  # noqa: E402
for kv in os.environ.get("RPT_OPTS", "").split(","):   # e.g. RPT_OPTS=walk_steal=0,detach_lanes=32
    if kv:
        rpt_amd.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if os.environ.get("CHUNK_SPP"):
    rpt_amd.set_option("chunk_spp", int(os.environ["CHUNK_SPP"]))
rpt_amd.set_option("counters", 1)
r.sample_array(spp)
c = r.counters()
out = (C.c_uint64 * 56)()
_lib.check(_lib.load().rpt_debug_section_counters(r.scene._handle, out))
print(f"{name} {spp} spp: wave trips {c['wave_trips']}, rays/sample {c['rays'] / c['samples']:.3f}, tree nodes/ray {c['bvh_nodes'] / c['rays']:.2f}, "
      f"trips per 64 vertices {c['wave_trips'] * 64 / c['vertices']:.2f}")
for k, nm in enumerate(NAMES):
    w, l = int(out[2 * k]), int(out[2 * k + 1])
    if w:
        print(f"  {nm:48s} execs/trip {w / c['wave_trips']:.3f}   lanes {l / w:5.1f} / 64")
if int(out[53]):   # counters[61]
    print(f"  tree nodes visited by parked PRIMARY walks: {int(out[53]) / max(c['bvh_nodes'], 1):.3f} of all")
if int(out[38]):
    nd, tr = int(out[38]), int(out[39])   # wave-level steps of the deferred walks: descent, triangles
    print(f"  deferred walks: {nd / c['wave_trips']:.2f} descent steps per trip with {c['bvh_nodes'] / nd:5.1f} / 64 lanes, "
          f"{tr / c['wave_trips']:.2f} triangle steps per trip with {c['bvh_tris'] / max(tr, 1):5.1f} / 64 lanes")
