import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes
sc, cam, cfg = scenes.CONFIGS["C3"]()
rpt_amd.set_option("timing", 1)
d_out = torch.zeros(1024 * 1024 * 3, dtype=torch.float64, device="cuda")
for chunk in (4, 2):
    rpt_amd.set_option("chunk_spp", chunk)
    for count in (8, 1):
        res = []
        for rank in range(count):
            r = Renderer(sc, cam).width(1024).height(1024).max_bounces(10).seed(0).shard(rank, count)
            ms = []
            for i in range(3):
                r._sample_offset = 0
                r.sample_device(256, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                ms.append(r.timing()[0])
            res.append(min(ms[1:]))
        print("chunk", chunk, "shards", count, "per-rank ms", ["%.2f" % x for x in res], "max %.2f ideal %.2f eff %.2f" % (max(res), 32.7 / count, 32.7 / count / max(res)), flush=True)
