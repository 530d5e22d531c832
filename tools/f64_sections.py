"""Static instruction counts per marked section of render_f64_kernel (-DRPT_MARKERS build: "; SECT k" comments in the ISA).
    python tools/f64_sections.py [mangled-name-fragment]      (CPU only; compiles kernels_f64.hip to assembly)
With --run (on the GPU box) the counters build's dynamic counts stand beside them: executions per loop trip and lanes enabled per
execution, for C3 (or the workload named) at 512 x 512 x 32 with the plain build's search limits (f64_cull = 2)."""
import collections
import re
import subprocess
import sys

NAMES = ["hand-out", "next sample", "distance sample", "query: setup + box tests", "chunk: deal pairs, fetch rays", "pair: record + transform",
         "plane", "sphere", "slab roots", "cube", "mesh bounds", "triangle: plane", "triangle: inside test", "collect one rank",
         "after a query", "medium event", "surface event", "shadow result", "light sample", "light term (medium)", "light term (surface)",
         "bounce (medium)", "bounce (surface)", "carrier update"]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
frag = "ILb1ELb0ELb1E"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm",
       "--cuda-device-only", "-S", "-DRPT_MARKERS", "rpt_amd/csrc/kernels_f64.hip", "-o", "/tmp/f64_marked.s"]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
inside, sect = False, None
cnt = collections.defaultdict(lambda: collections.Counter())
for line in open("/tmp/f64_marked.s"):
    t = line.strip()
    if t.startswith("_ZN5rpt6417render_f64_kernel") and ":" in t.split()[0]:
        inside = frag in t
        sect = "entry"
        continue
    if not inside:
        continue
    if t.startswith(".Lfunc_end"):
        inside = False
        continue
    m = re.match(r";+ *SECT (\d+)", t)
    if m:
        sect = int(m.group(1))
        continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    c = cnt[sect]
    c["all"] += 1
    if op.startswith("v_") and "f64" in op:
        c["f64"] += 1
    elif op.startswith("v_"):
        c["valu32"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        c["vmem"] += 1
dyn = None
if "--run" in sys.argv:
    import ctypes as C
    sys.path.insert(0, ".")
    from rpt_amd import Renderer, scenes, _lib
    name = args[0] if args else "C3"
    scene, cam, cfg = scenes.CONFIGS[name]()
    for k, v in (("epsilon_policy", 1), ("counters", 1), ("f64_cull", 2)):
        scene.set_option(k, v)
    r = Renderer(scene, cam).width(512).height(512).max_bounces(cfg["max_bounces"]).seed(0)
    r.sample_array(32)
    out = (C.c_uint64 * 12)()
    _lib.check(_lib.load().rpt_debug_epsilon_counters(scene._handle, out))
    sec = (C.c_uint64 * 56)()
    _lib.check(_lib.load().rpt_debug_section_counters(scene._handle, sec))
    trips = max(int(out[10]), 1)
    dyn = {k: (int(sec[2 * k]) / trips, int(sec[2 * k + 1]) / max(int(sec[2 * k]), 1)) for k in range(24)}
    print(f"{name} 512x512x32: {trips} wave trips, {int(out[11]) / trips:.1f} lanes with a path per trip, {int(out[9]) / trips:.2f} chunks per trip, "
          f"{int(out[8]) / max(int(out[0]), 1):.2f} objects evaluated per query")
print(f"{'section':>32} {'all':>6} {'f64':>6} {'valu32':>7} {'salu':>6} {'lds':>5} {'vmem':>5}" + ("   per trip   lanes   instr/trip" if dyn else ""))
tot, model = collections.Counter(), 0.0
for k in sorted(cnt, key=lambda x: (isinstance(x, str), x)):
    c = cnt[k]
    tot.update(c)
    label = NAMES[k] if isinstance(k, int) and k < len(NAMES) else str(k)
    line = f"{label:>32} {c['all']:6d} {c['f64']:6d} {c['valu32']:7d} {c['salu']:6d} {c['lds']:5d} {c['vmem']:5d}"
    if dyn and isinstance(k, int):
        e, l = dyn[k]
        model += e * c["all"]
        line += f"   {e:8.3f} {l:7.1f} {e * c['all']:10.0f}"
    print(line)
print(f"{'total':>32} {tot['all']:6d} {tot['f64']:6d} {tot['valu32']:7d} {tot['salu']:6d} {tot['lds']:5d} {tot['vmem']:5d}" + (f"   modelled instructions per trip {model:.0f}" if dyn else ""))
