import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import rpt_amd
from rpt_amd import Renderer, scenes, _lib
sc, cam, cfg = scenes.CONFIGS["C3"]()
rpt_amd.set_option("counters", 1)
for count in (1, 8):
    r = Renderer(sc, cam).width(1024).height(1024).max_bounces(10).seed(0).shard(0, count)
    r.sample_array(256)
    out = (C.c_uint64 * 56)()
    _lib.check(_lib.load().rpt_debug_trip_stamps(r.scene._handle, out))
    st = np.array([int(v) for v in out], dtype=np.float64)
    st = st[st > 0]
    d = np.diff(st) / 100.0 / 32.0   # us per trip
    print("shards", count, "us/trip over successive 32-trip windows:", np.round(d[:40], 1))
