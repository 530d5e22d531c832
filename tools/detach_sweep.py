"""Per-mesh-tree scenes in a medium: kernel time with the shadow queries detached (wave queue in LDS) against the parked
form, over "detach_lanes" (parked primary + queued shadow queries that trigger a walk session), "detach_trigger" (queued
shadow queries alone) and "defer_stop".  All detached frames must be bit-identical; the parked frame differs in the last bits
(another order of the same sums).
A value that starts with "s" is the streamed form (detach_shadows = 2: primary queries leave as well, their paths wait in
memory): s:backlog:stop[:contexts].
Usage: python tools/detach_sweep.py [workload] [width] [spp] [lanes:trigger:stop[:walk_leaf_quarters] | s:backlog:stop[:q] ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
values = [tuple(x if x == "s" else int(x) for x in v.split(":")) for v in sys.argv[4:]] or [(48, 20, 16), (32, 20, 16), (40, 20, 16), (56, 24, 16), (64, 28, 16), (48, 20, 8),
                                                                          (48, 20, 24), (48, 12, 16), (48, 28, 16), (64, 32, 24)]
scene, cam, cfg = scenes.CONFIGS[name]()
rpt_amd.set_option("timing", 1)
if os.environ.get("CHUNK_SPP"):   # samples per work item (the automatic rule follows the sample count: 16 at C5's 1024 spp, 2 at 64)
    rpt_amd.set_option("chunk_spp", int(os.environ["CHUNK_SPP"]))


def run(label):
    r = Renderer(scene, cam).width(width).height(width).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(4)
    ms = []
    for _ in range(2):
        r._sample_offset = 0
        img = r.sample_array(spp)
        ms.append(r.timing()[0])
    print(f"{name} {width}x{width}x{spp} {label}: kernel {min(ms):9.3f} ms   mean {img.mean():.9f}   grid {r.timing()[2]} blocks", flush=True)
    return img


rpt_amd.set_option("detach_shadows", 0)
parked = run("parked shadow queries (32:16)      ")
ref = None
for v in values:
    if v[0] == "s":
        rpt_amd.set_option("detach_shadows", 2)
        rpt_amd.set_option("stream_backlog", v[1])
        rpt_amd.set_option("stream_contexts", v[3] if len(v) > 3 else 4)
        label = f"streamed backlog:stop:contexts={v[1]:3d}:{v[2]:2d}:{v[3] if len(v) > 3 else 4}"
        v = v[:3]
    else:
        rpt_amd.set_option("detach_shadows", 1)
        rpt_amd.set_option("detach_lanes", v[0])
        rpt_amd.set_option("detach_trigger", v[1])
        label = f"detached lanes:trigger:stop={v[0]:2d}:{v[1]:2d}:{v[2]:2d}"
    rpt_amd.set_option("defer_stop", v[2])
    rpt_amd.set_option("walk_leaf_quarters", v[3] if len(v) > 3 else 6)
    img = run(label + (f" leaf quarters {v[3]:2d}" if len(v) > 3 else "                 "))
    if ref is None:
        ref = img
        d = img - parked
        print(f"   vs parked: rel RMS {np.sqrt((d ** 2).mean() / (parked ** 2).mean()):.3e}, max abs {np.abs(d).max():.3e}, finite {np.isfinite(img).all()}", flush=True)
    else:
        print(f"   identical to the first detached frame: {np.array_equal(img, ref)}", flush=True)
