"""Kernel time of every rank's tile shard of C3 for n = 2, 4, 8 (ranks emulated one after the other on
one GPU): the slowest rank sets the multi-GPU step time.  Usage: python tools/shard_ranks.py"""
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402
import rpt_amd  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

sc, cam, cfg = scenes.CONFIGS["C3"]()
rpt_amd.set_option("timing", 1)
d_out = torch.zeros(1024 * 1024 * 3, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
full = None
for count in (1, 2, 4, 8):
    times = []
    for rank in range(count):
        r = Renderer(sc, cam).width(1024).height(1024).max_bounces(10).seed(0).shard(rank, count)
        best = 1e9
        for i in range(3):
            r._sample_offset = 0
            r.sample_device(256, d_out.data_ptr(), st)
            torch.cuda.synchronize()
            best = min(best, r.timing()[0] + r.timing()[1])
        times.append(best)
    if full is None:
        full = times[0]
    print(f"n={count}: per-rank kernel+resolve ms {[round(t, 2) for t in times]}  max {max(times):.2f}  "
          f"ideal {full / count:.2f}  efficiency {full / count / max(times):.3f}", flush=True)
