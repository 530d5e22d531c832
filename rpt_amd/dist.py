"""One-process-per-GPU plumbing.  The path has exactly two exchange steps, and both are in the
library, behind the C ABI, over a communicator of its own (`FrameComm` = rpt_comm_*: RCCL bound with dlopen):

  * frames: every rank renders its 32x32 tile shard; the owned tiles travel packed (f64) to rank 0, which scatters them
    into its frame (`FrameComm.gather` = rpt_gather_frame_device: ncclSend / grouped ncclRecv);
  * photon maps: the shooting loop (src/photon.rs:656-690) is sharded by photon index, every rank needs the whole map,
    so the shot records are all-gathered in rank order (`FrameComm.allgather_records` = rpt_allgather_records_device;
    `photon_map_build_sharded`).

torch.distributed is the launcher and the store that hands out the communicator id (backend "nccl" = RCCL on ROCm, "gloo"
in the CPU tests).  `reduce_frame` (sum-reduce of zero-padded full frames) and `gather_records` are the same two steps
through torch.distributed itself: what the CPU tests run with gloo, and the fallback when a rank cannot form the
library's communicator.

torch is imported lazily so that `import rpt_amd` stays numpy-only."""
import ctypes as _C

from . import _lib

COMM_ID_BYTES = 128  # RPT_COMM_ID_BYTES
GATHER_LOOPBACK = 1  # RPT_GATHER_LOOPBACK


class FrameComm:
    """rpt_comm: an RCCL communicator of the library's own, one rank per GPU.

    FrameComm.unique_id() on rank 0 -> hand the 128 bytes to every rank -> FrameComm(id, rank, world, device)
    on every rank (collective).  gather(width, height, d_shard, d_frame, stream) assembles the sharded frames on rank 0."""

    def __init__(self, uid, rank, world, device):
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("communicator id must be 128 bytes")
        buf = (_C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        h = _C.c_void_p()
        _lib.check(_lib.load().rpt_comm_create(_C.cast(buf, _C.c_void_p), int(rank), int(world), int(device), _C.byref(h)))
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id():
        buf = (_C.c_char * COMM_ID_BYTES)()
        _lib.check(_lib.load().rpt_comm_unique_id(_C.cast(buf, _C.c_void_p)))
        return bytes(buf)

    @classmethod
    def from_torch(cls, dist, device):
        """Create the communicator of an initialised torch.distributed job: rank 0 draws the id, the job's store /
        broadcast hands it out (any backend: the id is 128 bytes of host data)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        # Rank 0 broadcasts whatever happened: were it to raise before the broadcast (librccl.so.1 missing, say), the other
        # ranks would sit in theirs while rank 0 moves on to its next collective.
        box = [None]
        if rank == 0:
            try:
                box = [cls.unique_id()]
            except Exception as e:   # noqa: BLE001 -- handed to every rank below
                box = [f"{type(e).__name__}: {e}"]
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise _lib.RptError(f"rank 0 could not draw a communicator id ({box[0]})")
        return cls(box[0], rank, world, device)

    def gather(self, width, height, d_shard, d_frame=None, stream=None, loopback=False):
        _lib.check(_lib.load().rpt_gather_frame_device(self._h, int(width), int(height), _C.c_void_p(d_shard),
                                                       _C.c_void_p(d_frame) if d_frame else None,
                                                       GATHER_LOOPBACK if loopback else 0, _C.c_void_p(stream) if stream else None))

    def allgather_records(self, d_local, n_local, d_out, capacity, stream=None):
        """rpt_allgather_records_device: every rank's photon records in rank order -> (per-rank counts, total).
        With d_out = None and capacity = 0 only the counts are exchanged."""
        per = (_C.c_uint64 * self.world)()
        tot = _C.c_uint64()
        _lib.check(_lib.load().rpt_allgather_records_device(self._h, _C.c_void_p(d_local) if n_local else None, int(n_local),
                                                            _C.c_void_p(d_out) if d_out else None, int(capacity), per, _C.byref(tot),
                                                            _C.c_void_p(stream) if stream else None))
        return [int(v) for v in per], int(tot.value)

    def close(self):
        if self._h:
            _lib.load().rpt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_pack_layout(width, height, world):
    """rpt_frame_pack_layout: tile offsets of every rank's block in the gathered buffer (world + 1 values)."""
    out = (_C.c_uint64 * (world + 1))()
    _lib.check(_lib.load().rpt_frame_pack_layout(int(width), int(height), int(world), out))
    return [int(v) for v in out]

RECORD_BYTES = 48  # RPT_PHOTON_RECORD_BYTES


def reduce_frame(frame, group=None):
    """Sum the per-rank frames onto rank 0 (non-owned pixels are exact zeros, so the sum is the frame)."""
    import torch.distributed as dist
    dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM, group=group)
    return frame


def gather_records(local, group=None):
    """All-gather variable-length record arrays in rank order.

    local: (n, RECORD_BYTES) uint8 tensor on this rank's device.  Returns the (sum n, RECORD_BYTES)
    concatenation, identical on every rank.  Shards are contiguous photon blocks, so this IS the array a
    single GPU would have produced."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if local.dim() != 2 or local.shape[1] != RECORD_BYTES or local.dtype != torch.uint8:
        raise ValueError("records must be an (n, 48) uint8 tensor")
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if world == 1:
        return local
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty((cap, RECORD_BYTES), dtype=torch.uint8, device=local.device) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][:counts[r]] for r in range(world)], dim=0).contiguous()


class _DevicePtr:
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned device memory."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _view(ptr, n, device):
    import torch
    if n == 0:
        return torch.empty((0, RECORD_BYTES), dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevicePtr(ptr, n * RECORD_BYTES), device=device).view(n, RECORD_BYTES)


def photon_map_build_sharded(renderer, photon_count, kind, rank, world, group=None, comm=None):
    """Renderer.photon_map_build across `world` ranks: each rank shoots its block of photons
    (rpt_photon_shoot), the records are all-gathered over RCCL -- through the library's own communicator
    (`comm`: a FrameComm, rpt_allgather_records_device) or through torch.distributed -- and every rank builds the
    full map (rpt_photon_map_from_records).  Returns the stats dict of photon_map_build."""
    import torch
    if renderer.scene._options.get("epsilon_policy", 0) == 1:
        # reference-epsilon mode: the visibility rays start at the photons' fp64 positions, which the 48-byte records do not carry --
        # every rank shoots the whole map itself (same seed, same streams: the same map everywhere; the camera pass still shards)
        return renderer.photon_map_build(photon_count, kind)
    device = torch.device("cuda", renderer.device_)
    renderer.photon_shoot(photon_count, kind, rank, world)
    gathered = []
    for which in (0, 1):
        ptr, n = renderer.photon_records(which)
        if comm is not None:
            # the counts first (a photon stores one record per scattering event: no bound short of the walk's length), then
            # the records into a buffer of exactly that size
            st = torch.cuda.current_stream(device).cuda_stream
            _, total = comm.allgather_records(ptr, n, None, 0, st)
            out = torch.empty((max(total, 1), RECORD_BYTES), dtype=torch.uint8, device=device)
            _, total = comm.allgather_records(ptr, n, out.data_ptr(), max(total, 1), st)
            gathered.append(out[:total])
            continue
        gathered.append(gather_records(_view(ptr, n, device), group))
    torch.cuda.current_stream(device).synchronize()   # the library builds on the null stream
    return renderer.photon_map_from_records(photon_count, kind, gathered[0].data_ptr(), gathered[0].shape[0],
                                            gathered[1].data_ptr(), gathered[1].shape[0])


__all__ = ["RECORD_BYTES", "reduce_frame", "gather_records", "photon_map_build_sharded", "FrameComm", "frame_pack_layout"]
