"""Frame exchange behind the C ABI (include/rpt_hip.h "frame exchange", rpt_amd/csrc/rpt_comm.cpp): the pack / unpack of
owned tiles on the device, and the RCCL transport with the one rank a single-GPU box has (rank 0 sends its own tiles
to itself through ncclSend / ncclRecv).  The N > 1 arithmetic is the same tile lists (CPU test in test_dist_gloo.py)."""
import ctypes as C

import numpy as np
import pytest

from rpt_amd import _lib, shard_pixels
from rpt_amd.dist import FrameComm, frame_pack_layout

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("w,h,n", [(256, 192, 3), (100, 70, 2), (64, 64, 1), (33, 65, 8)])
def test_pack_and_unpack_move_exactly_the_owned_tiles(w, h, n):
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(w * 7 + n)
    frame_h = rng.normal(size=(h * w, 3))
    frame = torch.from_numpy(frame_h.reshape(-1)).cuda()
    offs = frame_pack_layout(w, h, n)
    total = torch.zeros(h * w * 3, dtype=torch.float64, device="cuda")
    for r in range(n):
        tiles = offs[r + 1] - offs[r]
        packed = torch.full((max(tiles, 1) * 3072,), 7.0, dtype=torch.float64, device="cuda")
        _lib.check(lib.rpt_frame_pack_device(w, h, r, n, _ptr(frame), _ptr(packed), None))
        one = torch.zeros(h * w * 3, dtype=torch.float64, device="cuda")
        _lib.check(lib.rpt_frame_unpack_device(w, h, r, n, _ptr(packed), _ptr(one), None))
        torch.cuda.synchronize()
        got = one.cpu().numpy().reshape(-1, 3)
        own = shard_pixels(w, h, r, n)
        exp = np.zeros_like(frame_h)
        exp[own] = frame_h[own]
        assert np.array_equal(got, exp)                        # owned pixels bit for bit, nothing else touched
        if tiles:                                              # out-of-image slots of clipped tiles are zero, not stale
            assert float(packed[:tiles * 3072].abs().sum()) == pytest.approx(float(np.abs(frame_h[own]).sum()), rel=1e-12)
        total += one
    assert np.array_equal(total.cpu().numpy().reshape(-1, 3), frame_h)


def test_one_rank_gather_through_rccl():
    """ncclCommInitRank with one rank, then the gather with RPT_GATHER_LOOPBACK: rank 0's tiles are packed, sent to itself
    with ncclSend / ncclRecv inside one group, and unpacked into a second frame."""
    import torch
    w, h = 200, 136
    comm = FrameComm(FrameComm.unique_id(), 0, 1, 0)
    try:
        rng = np.random.default_rng(5)
        src_h = rng.normal(size=h * w * 3)
        src = torch.from_numpy(src_h).cuda()
        dst = torch.zeros_like(src)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            comm.gather(w, h, src.data_ptr(), dst.data_ptr(), st.cuda_stream, loopback=True)
        st.synchronize()
        assert np.array_equal(dst.cpu().numpy(), src_h)
        # without the loopback flag the own tiles are copied on the device; in place it is a no-op
        dst2 = torch.zeros_like(src)
        comm.gather(w, h, src.data_ptr(), dst2.data_ptr(), None)
        comm.gather(w, h, src.data_ptr(), src.data_ptr(), None)
        torch.cuda.synchronize()
        assert np.array_equal(dst2.cpu().numpy(), src_h) and np.array_equal(src.cpu().numpy(), src_h)
        # another frame size on the same communicator
        src3 = torch.from_numpy(rng.normal(size=64 * 32 * 3)).cuda()
        dst3 = torch.zeros_like(src3)
        comm.gather(64, 32, src3.data_ptr(), dst3.data_ptr(), None, loopback=True)
        torch.cuda.synchronize()
        assert torch.equal(src3, dst3)
    finally:
        comm.close()


def test_one_rank_all_gather_of_photon_records():
    """rpt_allgather_records_device with the one rank of a single-GPU box: counts, order and bytes of the records; an output
    that is too small is refused with the total reported."""
    import torch
    from rpt_amd import RptError
    comm = FrameComm(FrameComm.unique_id(), 0, 1, 0)
    try:
        rec = torch.randint(0, 256, (1234, 48), dtype=torch.uint8, device="cuda")
        out = torch.zeros((2000, 48), dtype=torch.uint8, device="cuda")
        per, total = comm.allgather_records(rec.data_ptr(), rec.shape[0], out.data_ptr(), out.shape[0])
        torch.cuda.synchronize()
        assert per == [1234] and total == 1234 and torch.equal(out[:1234], rec) and int(out[1234:].sum()) == 0
        per, total = comm.allgather_records(0, 0, out.data_ptr(), out.shape[0])      # a rank that shot nothing
        assert per == [0] and total == 0
        with pytest.raises(RptError):
            comm.allgather_records(rec.data_ptr(), rec.shape[0], out.data_ptr(), 10)
    finally:
        comm.close()


def test_counts_only_call_and_a_medium_that_stores_many_records_per_photon():
    """rpt_allgather_records_device with d_out = NULL, capacity = 0 exchanges the counts only; the sharded map build sizes
    its buffers from it.  In a dense, nearly non-absorbing medium a photon stores far more than four records (the walk goes
    on with probability sigma_s / sigma_t): the comm path must build the same map as the single-process call."""
    import torch
    from rpt_amd import Renderer, scenes
    from rpt_amd.dist import photon_map_build_sharded
    comm = FrameComm(FrameComm.unique_id(), 0, 1, 0)
    try:
        rec = torch.randint(0, 256, (77, 48), dtype=torch.uint8, device="cuda")
        per, total = comm.allgather_records(rec.data_ptr(), 77, None, 0)
        assert per == [77] and total == 77
        stats = []
        for use_comm in (False, True):
            scene, cam, cfg = scenes.lampshade(absorb=0.0002, scat=0.02)   # albedo 0.99, mean free path 50 in a 550-unit room
            r = Renderer(scene, cam).width(32).height(32).gather_size(20).gather_size_volume(3).watts(1000.0)
            if use_comm:
                stats.append(photon_map_build_sharded(r, 5000, Renderer.PHOTON_POINT_BEAM, 0, 1, comm=comm))
            else:
                stats.append(r.photon_map_build(5000, Renderer.PHOTON_POINT_BEAM))
        for k in ("surface", "volume", "shot"):
            assert stats[0][k] == stats[1][k], k
        assert stats[0]["volume"] > 4 * 5000     # (what a buffer of 4 x photon_count records could not hold)
    finally:
        comm.close()


def test_two_ranks_assemble_the_single_gpu_frame_and_map():
    """Two processes, two GPUs (skipped on a one-GPU box): the gathered frame and the all-gathered photon records equal the
    single-GPU arrays bit for bit -- rank 0's staging offsets, the one unpack over the other ranks' blocks, the non-root
    ncclSend and the padded all-gather with unequal counts all take part.  The ranks are started before anything touches a
    GPU in this process's children (tests/mp_comm_worker.py under torch.distributed.run)."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "tests", "mp_comm_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "frames equal" in out.stdout and "records equal" in out.stdout
