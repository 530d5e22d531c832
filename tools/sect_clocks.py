"""Where a wave of the megakernel spends its time: wave clock ticks per section of the loop body.
Needs a diagnostic build of the library with -DRPT_SECT_CLOCKS (the lanes slot of each section counter then holds ticks):
  cd rpt_amd/csrc && hipcc <__graft_entry__.FLAGS> -DRPT_SECT_CLOCKS -o ../librpt_hip_clk.so <SRCS> -ldl
  RPT_LIB=$PWD/rpt_amd/librpt_hip_clk.so python tools/sect_clocks.py [workload] [spp] [width]
A section's ticks run from the moment the wave reaches that section point until it reaches the next one, stalls included
(only the points inside render_kernel's own body take part; the stage functions' sections 5, 6, 10-13 do not)."""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, _lib, scenes  # noqa: E402

NAMES = {0: "work pull", 1: "regenerate camera ray", 2: "vertex start: distance sample + primary query", 3: "after primary query: event, finalize, material",
         4: "miss / environment", 7: "light sample", 8: "shadow query", 9: "after shadow query: light term, bounce",
         14: "path update (until the next trip)", 15: "parked tree walks", 16: "walk finished", 17: "... with a triangle hit",
         18: "... shadow query", 21: "detached: shadow query queued", 22: "detached: queue full, walked in place", 27: "before the first section point"}
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene, cam, cfg = scenes.CONFIGS[name]()
width = int(sys.argv[3]) if len(sys.argv) > 3 else min(cfg["width"], 1024)
r = Renderer(scene, cam).width(width).height(width * cfg["height"] // cfg["width"]).max_bounces(cfg["max_bounces"]).seed(0)
if os.environ.get("CHUNK_SPP"):
    rpt_amd.set_option("chunk_spp", int(os.environ["CHUNK_SPP"]))
rpt_amd.set_option("counters", 1)
r.sample_array(spp)
c = r.counters()
out = (C.c_uint64 * 56)()
_lib.check(_lib.load().rpt_debug_section_counters(r.scene._handle, out))
total = sum(int(out[2 * k + 1]) for k in range(28))
print(f"{name} {width} x {spp} spp: wave trips {c['wave_trips']}, total wave ticks {total:.3e} ({total / max(c['wave_trips'], 1):.0f} per trip)")
for k in range(28):
    w, t = int(out[2 * k]), int(out[2 * k + 1])
    if w or t:
        print(f"  {k:2d} {NAMES.get(k, ''):52s} reached {w / max(c['wave_trips'], 1):6.3f} / trip   ticks {100.0 * t / total:5.1f} %   {t / max(w, 1):8.0f} per visit")
