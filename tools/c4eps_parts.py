"""C4 in the reference-epsilon mode at its configured size: the camera pass with parts switched off (option "photon_skip":
1 = no volume estimate, 2 = no surface estimate, 4096 = no visibility rays)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
from rpt_amd import Renderer, scenes  # noqa: E402

scene, cam, cfg = scenes.CONFIGS["C4"]()
scene.set_option("epsilon_policy", 1)
for kv in sys.argv[1:]:
    scene.set_option(kv.split("=")[0], int(kv.split("=")[1]))
n = cfg["photons"]
r = Renderer(scene, cam).width(1024).height(1024).watts(cfg["renderer_watts"]).gather_size(cfg["gather_size"]).gather_size_volume(cfg["gather_size_volume"]).seed(0)
r.photon_map_build(n, 1)
frame = torch.zeros(1024 * 1024 * 3, dtype=torch.float64, device="cuda")
for skip in (0, 4096, 1, 1 + 4096, 2):
    scene.set_option("photon_skip", skip)
    ts = []
    for i in range(3):
        r._sample_offset = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.photon_sample_device(256, frame.data_ptr(), 0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"photon_skip = {skip:5d}: camera pass {min(ts[1:]) * 1e3:8.2f} ms   (first call {ts[0] * 1e3:.1f})")
