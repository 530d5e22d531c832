"""The N > 1 path on CPU: world_size-2 gloo ranks each produce their tile shard (zero elsewhere)
and a sum-reduce on rank 0 assembles the frame bit-exactly.  The shard pixels come from the same
rpt_shard_tiles the HIP renderer uses; the pixel values come from the oracle (no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import OracleScene
    from rpt_amd import scenes, shard_pixels
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene, cam, cfg = scenes.lampshade()
    pix = shard_pixels(w, h, rank, world)
    frame = OracleScene(scene).render(cam, w, h, spp, cfg["max_bounces"], seed=5, threads=2, pixels=pix)
    mine = np.zeros(w * h, dtype=bool)
    mine[pix] = True
    assert np.all(frame[~mine] == 0.0)                      # non-owned pixels contribute exact zeros
    t = torch.from_numpy(frame)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_shards_sum_to_the_full_frame(tmp_path):
    import torch.multiprocessing as mp
    from oracle.pyoracle import OracleScene
    from rpt_amd import scenes
    w, h, spp = 96, 64, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, spp, out), nprocs=2, join=True)
    got = np.load(out)
    scene, cam, cfg = scenes.lampshade()
    full = OracleScene(scene).render(cam, w, h, spp, cfg["max_bounces"], seed=5, threads=2)
    assert np.array_equal(got, full)                        # bit-identical: RNG is keyed by (seed, pixel, sample)
    assert full.max() > 0


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from rpt_amd.dist import RECORD_BYTES, gather_records
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    for case, counts in enumerate([(5, 3), (0, 4), (7, 0), (0, 0)]):          # ragged and empty shards
        g = torch.Generator().manual_seed(100 * case + rank)
        local = torch.randint(0, 256, (counts[rank], RECORD_BYTES), dtype=torch.uint8, generator=g)
        got = gather_records(local)
        torch.save((local, got), os.path.join(out_dir, f"c{case}_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_photon_records_all_gather_in_rank_order(tmp_path):
    """The photon-map exchange step (rpt_amd.dist.gather_records): variable-length shards are
    concatenated in rank order on every rank, i.e. the single-GPU record array."""
    import torch
    import torch.multiprocessing as mp
    mp.spawn(_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for case in range(4):
        l0, g0 = torch.load(tmp_path / f"c{case}_r0.pt")
        l1, g1 = torch.load(tmp_path / f"c{case}_r1.pt")
        assert torch.equal(g0, g1) and torch.equal(g0, torch.cat([l0, l1], dim=0))
