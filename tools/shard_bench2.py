"""Kernel time of one tile shard (rank 0 of `count`) of C3: single launches separated by host
synchronisation vs back-to-back launches (does the idle gap cost clock ramp-up?)."""
import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes
sc, cam, cfg = scenes.CONFIGS["C3"]()
rpt_amd.set_option("timing", 1)
d_out = torch.zeros(1024 * 1024 * 3, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for count in (8, 4, 2, 1):
    r = Renderer(sc, cam).width(1024).height(1024).max_bounces(10).seed(0).shard(0, count)
    r.sample_device(256, d_out.data_ptr(), st); torch.cuda.synchronize()
    single = []
    for i in range(4):
        r._sample_offset = 0
        r.sample_device(256, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        single.append(r.timing()[0])
        time.sleep(0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 16
    e0.record()
    for i in range(n):
        r._sample_offset = 0
        r.sample_device(256, d_out.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    print("shards %d: single-launch kernel ms %s | back-to-back per-launch ms %.3f (incl. resolve+memsets) | ideal %.3f" % (count, ["%.2f" % x for x in single], e0.elapsed_time(e1) / n, 32.7 / count), flush=True)
