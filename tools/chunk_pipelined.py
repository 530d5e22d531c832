"""Step time of C3 against samples per work item ("chunk_spp"), as bench.py measures it: pipelined over two HIP streams, and one
step alone; for the whole frame and for one rank's shard of an 8-GPU job (--emulate-shard).  Each configuration is one
bench.py process.  Usage: python tools/chunk_pipelined.py [chunks...]"""
import json
import subprocess
import sys

chunks = [int(v) for v in sys.argv[1:]] or [4, 8, 16, 32]
for shard in (0, 8):
    for c in chunks:
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "4", "--no-cpu-baseline", "--no-secondary",
                              "--chunk-spp", str(c), "--emulate-shard", str(shard)], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("failed:", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        print(f"shard {'1/' + str(shard) if shard else 'whole frame'} chunk_spp {c:2d}: pipelined {d['ms_per_step']:8.3f} ms/step, alone "
              f"{d['wall_clock_s'] * 1e3:8.3f} ms, kernel {d['roofline']['kernel_ms']:8.3f} ms", flush=True)
