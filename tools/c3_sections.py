"""Static instruction counts per marked section of the scan megakernel (-DRPT_MARKERS: "; SECT k" comments in the ISA), by class.
    python tools/c3_sections.py [mangled-name-fragment]        (CPU only)
Dynamic execution counts per section: tools/trips.py (GPU)."""
import collections
import re
import subprocess
import sys

frag = sys.argv[1] if len(sys.argv) > 1 else "ILb1ELi0ELb0ELb0ELi0E"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                       "-DRPT_MARKERS", "rpt_amd/csrc/kernels.hip", "-o", "/tmp/k32_marked.s"], stderr=subprocess.DEVNULL)
inside, sect = False, "entry"
cnt = collections.defaultdict(collections.Counter)
for line in open("/tmp/k32_marked.s"):
    t = line.strip()
    if t.startswith("_ZN4rptg13render_kernel") and ":" in t.split()[0]:
        inside, sect = frag in t, "entry"
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
    for cls, pred in (("valu", op.startswith("v_")), ("salu", op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_load", "s_cbranch", "s_branch"))),
                      ("branch", op.startswith(("s_cbranch", "s_branch"))), ("waitcnt", op.startswith("s_waitcnt")), ("nop", op.startswith("s_nop")),
                      ("smem", op.startswith("s_load")), ("lds", op.startswith("ds_")), ("vmem", op.startswith(("global_", "scratch_", "buffer_", "flat_")))):
        if pred:
            c[cls] += 1
cols = ["all", "valu", "salu", "branch", "waitcnt", "nop", "smem", "lds", "vmem"]
print("section " + " ".join(f"{c:>8}" for c in cols))
for k in sorted(cnt, key=lambda x: (isinstance(x, str), x)):
    print(f"{str(k):>7} " + " ".join(f"{cnt[k][c]:8d}" for c in cols))
