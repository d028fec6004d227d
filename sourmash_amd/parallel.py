"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Only the two places where the path has a real exchange use a collective:

  compare (BASELINE config C4)  every rank holds the whole CSR (400 MB at N = 10,000); the N x N pair
      matrix is cut into 16-row tiles dealt round-robin to the ranks, each rank computes its tiles on or
      above the diagonal (cyclic dealing balances the triangular work), ONE all-gather assembles the
      u32 count matrix (u32 instead of f64 halves the bytes on the per-link-bound xGMI ring), then every
      rank mirrors the triangle and converts to f64 Jaccard locally.

  gather (BASELINE config C5)   the database is sharded by dataset; query, counters and the min-set-cover
      loop are replicated.  Per round: local arg-max packed as (count << 32) | ~global_index, ONE u64
      MAX all-reduce (this also implements the reference's tie-break: highest count, then lowest index),
      the owner broadcasts the winning sketch (<= a few thousand u64), every rank intersects it with its
      copy of the query and updates its own counters.  The stop test is deterministic and replicated.

The numerical work is behind a small `backend` interface.  DeviceBackend (the product) drives the HIP
kernels through the raw C-ABI on torch CUDA tensors.  The distributed control flow itself is
backend-agnostic, which is what lets tests/test_parallel_gloo.py run these very functions with
world_size 2 over gloo on CPU tensors (there the test injects an oracle-backed backend).
"""
import ctypes as C

TILE = 16   # rows per compare tile (csrc/compare.hip CT)


def _dist():
    import torch.distributed as dist
    return dist


def world_info(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def tiles_for_rank(n, world, rank):
    "-> (first_tile, tile_stride, tile_count): 16-row tiles rank, rank + world, ... of an n-row problem"
    n_tiles = (n + TILE - 1) // TILE
    count = (n_tiles - rank + world - 1) // world if n_tiles > rank else 0
    return rank, world, count


class DeviceBackend:
    "HIP kernels through the raw C-ABI (smgpu_*_raw) on torch CUDA tensors."

    def __init__(self, device=None):
        import torch
        from ._lowlevel import lib
        from .utils import rustcall
        self.torch, self.lib, self.rustcall = torch, lib, rustcall
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._ws = None
        self._index, self._index_key = None, None

    # -- helpers --
    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def _s(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=dtype, device=self.device)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = self.torch.empty(int(nbytes * 1.25) + 1024, dtype=self.torch.uint8, device=self.device)
        return self._ws

    # -- compare --
    def compare_tiles(self, hashes, offsets, n, first, stride, count, method="auto"):
        """counts for the owned 16-row tiles.  method: "merge" (LDS-tiled merge walk, upper triangle),
        "bits" (bit rows + popcount, all columns) or "auto" (bits when the collection is dense enough)."""
        out = self.zeros((count * TILE, n), self.torch.int32)
        index = None
        if method in ("auto", "bits"):
            key = (hashes.data_ptr(), offsets.data_ptr(), n)
            if self._index_key != key:
                from .device import BitIndex
                self._index, self._index_key = BitIndex.build(hashes, offsets), key
            index = self._index
            if index is None and method == "bits":
                raise ValueError("collection too sparse for the bit-row path")
        if index is not None:
            index.compare_tiles(first, stride, count, out=out)
        else:
            self.rustcall(self.lib.smgpu_compare_blocks_raw, self._p(hashes), self._p(offsets), n, first, stride, count,
                          self._p(out), self._s())
        return out

    def symmetrize(self, common, n):
        self.rustcall(self.lib.smgpu_symmetrize_raw, self._p(common), n, self._s())

    def jaccard(self, common, offsets, n):
        out = self.empty((n, n), self.torch.float64)
        self.rustcall(self.lib.smgpu_jaccard_raw, self._p(common), self._p(offsets), n, 0, n, self._p(out), self._s())
        return out

    # -- gather --
    def overlaps(self, query, nq, hashes, offsets, ndb, counters, op):
        self.rustcall(self.lib.smgpu_overlap_raw, self._p(query), nq, self._p(hashes), self._p(offsets), ndb,
                      self._p(counters), op, self._s())

    def argmax(self, counters, ndb, index_base):
        best = self.zeros((1,), self.torch.int64)
        self.rustcall(self.lib.smgpu_argmax_raw, self._p(counters), ndb, index_base, self._p(best), self._s())
        return best

    def select(self, a, na, b, nb, invert):
        "sorted a ∩ b (invert=False) or a minus b (invert=True) -> (tensor, count)"
        out = self.empty((max(na, 1),), self.torch.int64)
        n_out = self.zeros((1,), self.torch.int64)
        ws = self._workspace(self.lib.smgpu_intersect_workspace_bytes(max(na, 1)))
        fn = self.lib.smgpu_subtract_raw if invert else self.lib.smgpu_intersect_raw
        self.rustcall(fn, self._p(a), na, self._p(b), nb, self._p(out), self._p(n_out), self._p(ws), ws.numel(), self._s())
        return out, int(n_out.item())


# ---------------------------------------------------------------------------------------------------
def compare_all_pairs_distributed(hashes, offsets, n, backend, group=None, want_jaccard=True):
    """N x N common-hash matrix (int32 bit patterns of u32) and f64 Jaccard on every rank.
    hashes / offsets: the full CSR, replicated on every rank."""
    dist = _dist()
    rank, world = world_info(group)
    first, stride, count = tiles_for_rank(n, world, rank)
    local = backend.compare_tiles(hashes, offsets, n, first, stride, count)
    n_tiles = (n + TILE - 1) // TILE
    if world == 1:
        full = local[:n]
    else:
        max_count = (n_tiles + world - 1) // world
        if local.shape[0] != max_count * TILE:                    # equal-sized pieces for all_gather
            pad = backend.zeros((max_count * TILE, n), local.dtype)
            pad[:local.shape[0]] = local
            local = pad
        pieces = [backend.empty((max_count * TILE, n), local.dtype) for _ in range(world)]
        dist.all_gather(pieces, local, group=group)               # the ONE collective of the compare path
        full = assemble_tiles(pieces, n, world, backend)
    backend.symmetrize(full, n)
    jac = backend.jaccard(full, offsets, n) if want_jaccard else None
    return full, jac


def assemble_tiles(pieces, n, world, backend):
    "un-deal the gathered shards: tile t lives in piece t % world at slot t // world -> [n][n]"
    n_tiles = (n + TILE - 1) // TILE
    max_count = (n_tiles + world - 1) // world
    full = backend.empty((n_tiles * TILE, n), pieces[0].dtype)
    view = full.view(n_tiles, TILE, n)
    for r in range(world):
        cnt = (n_tiles - r + world - 1) // world if n_tiles > r else 0
        if cnt:
            view[r::world] = pieces[r].view(-1, TILE, n)[:cnt]
    return full[:n].contiguous()


def pack_key(count, index):
    "(count << 32) | ~index as a non-negative python int (fits int64: counts < 2^31)"
    return (int(count) << 32) | (0xFFFFFFFF & ~int(index))


def unpack_key(key):
    key = int(key)
    return key >> 32, 0xFFFFFFFF & ~key


def gather_distributed(query, nq, shard_hashes, shard_offsets, n_shard, index_base, threshold_bp, scaled, backend,
                       group=None, max_rounds=None):
    """Min-set-cover gather over a dataset-sharded database.

    query: sorted u64 hashes (int64 bit patterns) replicated on every rank; shard_*: this rank's CSR of
    n_shard sketches whose global indices start at index_base.  Returns [(global index, |intersect|)],
    identical on every rank and identical to the single-process result (ties -> lowest global index)."""
    dist = _dist()
    rank, world = world_info(group)
    torch = backend.torch if hasattr(backend, "torch") else __import__("torch")
    counters = backend.zeros((max(n_shard, 1),), torch.int64)
    backend.overlaps(query, nq, shard_hashes, shard_offsets, n_shard, counters, 0)        # CounterGather.add
    # one-off layout exchange: who owns which global indices, and the longest sketch anywhere
    host_off = shard_offsets.cpu()
    longest = int((host_off[1:] - host_off[:-1]).max().item()) if n_shard else 0
    layout = backend.zeros((3,), torch.int64)
    layout[0], layout[1], layout[2] = index_base, n_shard, longest
    if world > 1:
        layouts = [backend.zeros((3,), torch.int64) for _ in range(world)]
        dist.all_gather(layouts, layout, group=group)
        layouts = [tuple(int(v) for v in t.tolist()) for t in layouts]
    else:
        layouts = [(index_base, n_shard, longest)]
    max_len = max(t[2] for t in layouts)
    results = []
    cur, ncur = query, nq
    while max_rounds is None or len(results) < max_rounds:
        if ncur == 0:
            break
        n_threshold_hashes = 0.0
        if threshold_bp:                                           # search.py:15-37, float arithmetic as in Python
            n_threshold_hashes = float(threshold_bp) / scaled
            if n_threshold_hashes / ncur > 1.0:
                break
        best = backend.argmax(counters, n_shard, index_base)       # local winner, packed with the tie-break
        if world > 1:
            dist.all_reduce(best, op=dist.ReduceOp.MAX, group=group)   # collective 1 of the round: 8 bytes
        key = int(best.item())
        if key == 0:
            break
        count, gidx = unpack_key(key)
        if count < n_threshold_hashes:
            break
        # collective 2 of the round: the owner broadcasts [len, hashes...] of the winning sketch
        owner = next(r for r, (base, cnt, _) in enumerate(layouts) if base <= gidx < base + cnt)
        msg = backend.zeros((1 + max(max_len, 1),), torch.int64)
        if owner == rank:
            lo, hi = int(host_off[gidx - index_base].item()), int(host_off[gidx - index_base + 1].item())
            msg[0] = hi - lo
            msg[1:1 + hi - lo] = shard_hashes[lo:hi]
        if world > 1:
            src = dist.get_global_rank(group, owner) if group is not None else owner
            dist.broadcast(msg, src=src, group=group)
        m = int(msg[0].item())
        row = msg[1:1 + max(m, 1)]
        isect, ni = backend.select(cur, ncur, row, m, invert=False)    # I = Q ∩ match
        results.append((gidx, ni))
        backend.overlaps(isect, ni, shard_hashes, shard_offsets, n_shard, counters, 1)   # consume
        cur, ncur = backend.select(cur, ncur, row, m, invert=True)     # Q <- Q minus the whole match
    return results
