"""One rank of tests/test_gpu_comm.py::test_two_ranks_assemble_the_single_gpu_frame_and_map (run under torch.distributed.run).
Rank 0 also renders the whole frame and shoots all photons alone, and compares."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpt_amd import Renderer, scenes  # noqa: E402
from rpt_amd.dist import FrameComm, RECORD_BYTES  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
comm = FrameComm.from_torch(dist, local)
w, h, spp = 200, 136, 8
scene, cam, cfg = scenes.lampshade()
r = Renderer(scene, cam).width(w).height(h).max_bounces(cfg["max_bounces"]).seed(4).device(local).shard(rank, world)
frame = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
r.sample_device(spp, frame.data_ptr(), 0)
torch.cuda.synchronize()
comm.gather(w, h, frame.data_ptr(), frame.data_ptr(), None)
torch.cuda.synchronize()
if rank == 0:
    scene1, cam1, _ = scenes.lampshade()
    whole = Renderer(scene1, cam1).width(w).height(h).max_bounces(cfg["max_bounces"]).seed(4).device(local).sample_array(spp)
    assert np.array_equal(frame.cpu().numpy().reshape(-1, 3), whole), "gathered frame differs from the single-GPU frame"
    print("frames equal", flush=True)

# photon records: contiguous blocks of the shooting loop in rank order are the single-GPU arrays
n_photons = 3001   # (unequal blocks)
r.gather_size(20).gather_size_volume(3).watts(1000.0)
r.photon_shoot(n_photons, Renderer.PHOTON_POINT_BEAM, rank, world)
gathered = []
for which in (0, 1):
    ptr, n = r.photon_records(which)
    _, total = comm.allgather_records(ptr, n, None, 0)
    out = torch.zeros((max(total, 1), RECORD_BYTES), dtype=torch.uint8, device="cuda")
    comm.allgather_records(ptr, n, out.data_ptr(), max(total, 1))
    torch.cuda.synchronize()
    gathered.append(out[:total].cpu().numpy())
if rank == 0:
    scene1, cam1, _ = scenes.lampshade()
    r1 = Renderer(scene1, cam1).width(w).height(h).device(local).gather_size(20).gather_size_volume(3).watts(1000.0)
    r1.photon_shoot(n_photons, Renderer.PHOTON_POINT_BEAM, 0, 1)
    for which in (0, 1):
        ptr, n = r1.photon_records(which)
        from rpt_amd.dist import _view
        alone = _view(ptr, n, torch.device("cuda", local)).cpu().numpy()
        assert np.array_equal(alone, gathered[which]), "all-gathered records differ from the single-GPU records"
    print("records equal", flush=True)
dist.barrier()
comm.close()
dist.destroy_process_group()
