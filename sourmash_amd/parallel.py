"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Only the two places where the path has a real exchange use a collective:

  compare (BASELINE config C4)  every rank holds the whole CSR (400 MB at N = 10,000); the N x N pair
      matrix is cut into 16-row tiles dealt round-robin to the ranks, each rank computes its tiles on or
      above the diagonal (cyclic dealing balances the triangular work), ONE all-gather assembles the
      u32 count matrix (u32 instead of f64 halves the bytes on the per-link-bound xGMI ring), then every
      rank mirrors the triangle and converts to f64 Jaccard locally.

  gather (BASELINE config C5)   the database is sharded by dataset; the query, its uncovered set and the
      min-set-cover loop are replicated.  Rounds are replayed from exchanged candidates: every rank exports its K
      best rows -- key (count << 32) | ~global_index (the reference's order: highest count, then lowest index),
      hashes, and the best key it keeps back -- ONE all-gather hands the W x K records to everyone, every rank
      inverts them against the query next to its own shard, and then rounds run back to back with no exchange: the
      best candidate is the global arg-max as long as its key is not below any kept-back key (counters only
      decrease), its row is already on every rank, and applying it updates the local counters through the shard's
      postings and the candidates' counters through their bit masks.  When the test fails the remaining rounds of the
      batch are no-ops and the next exchange decides.  The stop test is deterministic and replicated; nothing between
      two polls of the done flag needs the host.  (The per-round form -- an 8-byte MAX all-reduce plus a row
      broadcast each round -- would spend two collectives where a single-GPU round takes 25 us.)

The numerical work is behind a small `backend` interface.  DeviceBackend (the product) drives the HIP
kernels through the raw C-ABI on torch CUDA tensors.  The distributed control flow itself is
backend-agnostic, which is what lets tests/test_parallel_gloo.py run these very functions with
world_size 2 over gloo on CPU tensors (there the test injects an oracle-backed backend).
"""
import ctypes as C
import math

TILE = 16   # rows per compare tile (csrc/compare.hip CT)
COMPARE_BANDS = 4   # pieces the compare exchange is cut into so that it overlaps the tiles (compare_all_pairs_distributed)
COMPARE_BAND_MIN_SLOTS = 8   # ... when every piece holds at least this many 16-row tiles per rank
TOPK_MAX = 16    # candidates a rank can export per exchange (csrc/gather_api.hpp GATHER_TOPK_MAX)
CAND_MAX = 64    # candidates of all ranks together (GATHER_CAND_MAX: one mask bit each)
CAND_HEAD = 3    # record header: key, bound, len


def _dist():
    import torch.distributed as dist
    return dist


def world_info(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


# Collectives on whatever group the caller runs under.  The product configuration is one process per GPU under "nccl" (RCCL over
# xGMI) with device tensors.  A gloo group moves host memory only: device tensors are staged through the host around the
# collective -- what lets two real processes share ONE GPU in tests/test_gpu_two_processes.py and drive these very functions
# (DeviceBackend, kernels and all) where no second GPU exists.
def _staged(t, group):
    dist = _dist()
    return bool(t.is_cuda) and dist.get_backend(group) == "gloo"


def _all_reduce(t, op, group=None):
    dist = _dist()
    if _staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_gather(outs, t, group=None, async_op=False):
    "outs[r] <- rank r's t (same shape and dtype everywhere); async_op: -> a work handle to wait() on (None when done already)"
    dist = _dist()
    if _staged(t, group):
        c = t.cpu().contiguous()
        parts = [c.new_empty(c.shape) for _ in outs]
        dist.all_gather(parts, c, group=group)
        for o, part in zip(outs, parts):
            o.copy_(part)
        return None
    if async_op:
        return dist.all_gather(outs, t, group=group, async_op=True)
    dist.all_gather(outs, t, group=group)
    return None


def _gather_to_root(outs, t, group=None, async_op=False):
    """rank 0's outs[r] <- rank r's t (outs is None elsewhere): what the all-gather moves, to ONE receiver -- every rank sends its
    shard over its own link to rank 0 instead of every shard travelling round the ring"""
    dist = _dist()
    root = dist.get_global_rank(group, 0) if group is not None else 0
    if _staged(t, group):
        c = t.cpu().contiguous()
        parts = [c.new_empty(c.shape) for _ in outs] if outs is not None else None
        dist.gather(c, parts, dst=root, group=group)
        if outs is not None:
            for o, part in zip(outs, parts):
                o.copy_(part)
        return None
    if async_op:
        return dist.gather(t, outs, dst=root, group=group, async_op=True)
    dist.gather(t, outs, dst=root, group=group)
    return None


def agree(flag, group=None):
    "True iff `flag` holds on every rank (one MIN all-reduce; a plain bool without a process group)"
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    torch = __import__("torch")
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


EXCHANGE_RUNS_MAX = 4000       # run tags of the shared exchange carry 12 bits of the run number (gather.hip: epoch_base)


def open_shared_exchange(owner, make, world, rank, rowcap, group=None):
    """The exchange area of this job's ranks on one node, made once and reused while it is large enough: rank 0 creates a named
    segment (make(name, True)), the name travels by one broadcast, the others map it (make(name, False)).  Every step is AGREED
    before anybody goes on: a rank that failed leaves through the same collectives as its peers and all of them raise -- nobody
    is left inside a barrier the others never reach.  The area's tags tell runs apart by 12 bits of the run
    number, so it is made afresh before that wraps.  -> (exchange, run id); state lives on `owner` (_xchg, _xchg_runs)."""
    import os
    dist = _dist()
    cur = getattr(owner, "_xchg", None)
    grouped = world > 1 or (dist.is_available() and dist.is_initialized())
    if cur is None or cur.world != world or cur.rowcap < rowcap or getattr(owner, "_xchg_runs", 0) >= EXCHANGE_RUNS_MAX:
        owner._xchg = None
        if not grouped:
            owner._xchg = make(None, True)
        else:
            owner._xchg_seq = getattr(owner, "_xchg_seq", 0) + 1
            box = ["/smg_gx_%d_%d" % (os.getpid(), owner._xchg_seq)] if rank == 0 else [None]
            dist.broadcast_object_list(box, src=0, group=group)
            made, err = None, None
            if rank == 0:
                try:
                    made = make(box[0], True)                             # the segment exists and is zeroed
                except Exception as e:                                   # noqa: BLE001 -- whatever it is, the peers must learn of it
                    err = e
            if not agree(err is None, group):
                raise err or RuntimeError("the shared gather exchange could not be created on rank 0")
            if rank != 0:
                try:
                    made = make(box[0], False)
                except Exception as e:                                   # noqa: BLE001
                    err = e
            if not agree(err is None, group):                            # everybody has it mapped -- or nobody keeps it
                made = None
                raise err or RuntimeError("the shared gather exchange could not be mapped on some rank")
            owner._xchg = made
        owner._xchg_runs = 0
    owner._xchg_runs += 1
    return owner._xchg, owner._xchg_runs


def allgather_union(local_hashes, group=None, force=False, backend=None):
    """Sketching shards by records: every rank holds the sorted unique kept hashes of ITS records (int64 tensor of u64
    bit patterns); one all-gather later every rank holds the sketch of the whole input -- set union is associative,
    which is all `merge` (minhash.rs:432-516) needs for flat scaled sketches.  Two collectives: the sizes, then the
    padded hash vectors (~L / scaled / world u64 each); the union itself is the library's device sort + unique."""
    dist = _dist()
    rank, world = world_info(group)
    torch = __import__("torch")
    if world == 1 and not force:                        # force: run the collectives even alone (1-GPU test boxes)
        return local_hashes
    dev = local_hashes.device
    n_local = torch.tensor([local_hashes.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    _all_gather(sizes, n_local, group)
    sizes = [int(t.item()) for t in sizes]
    longest = max(max(sizes), 1)
    pad = torch.zeros(longest, dtype=torch.int64, device=dev)
    pad[:local_hashes.numel()] = local_hashes
    parts = [torch.empty(longest, dtype=torch.int64, device=dev) for _ in range(world)]
    _all_gather(parts, pad, group)
    merged = torch.cat([p[:n] for p, n in zip(parts, sizes)])
    if backend is not None:                              # (tests/test_parallel_gloo.py: the protocol over host tensors with its CPU stand-in)
        return backend.sort_unique(merged)
    from .device import sort_unique                      # the library's own radix sort + run-length encode (csrc/device_sort.hip)
    return sort_unique(merged)


def tiles_for_rank(n, world, rank):
    "-> (first_tile, tile_stride, tile_count): 16-row tiles rank, rank + world, ... of an n-row problem"
    n_tiles = (n + TILE - 1) // TILE
    count = (n_tiles - rank + world - 1) // world if n_tiles > rank else 0
    return rank, world, count


class _DeviceGatherState:
    "smgpu_gather_* over torch tensors: the per-rank state of a sharded gather (counters, postings, uncovered set)."

    def __init__(self, backend, query, nq, hashes, offsets, ndb, index_base):
        self.b, self.lib, self.rustcall = backend, backend.lib, backend.rustcall
        self._keep = (query, hashes, offsets)            # borrowed by the native object
        self._ptr = self.rustcall(self.lib.smgpu_gather_new_raw, backend._p(query), nq, backend._p(hashes),
                                  backend._p(offsets), ndb, index_base, backend._s())
        self._cap = 1

    def __del__(self):
        if getattr(self, "_ptr", None):
            self.lib.smgpu_gather_free(self._ptr)
            self._ptr = None

    def begin(self, threshold_hashes, max_rounds):
        self._cap = max(int(max_rounds), 1)
        self.rustcall(self.lib.smgpu_gather_begin, self._ptr, int(threshold_hashes), self._cap, self.b._s())

    def longest_row(self):
        return int(self.lib.smgpu_gather_longest_row(self._ptr))

    def export_topk(self, records, k):
        "this shard's k best rows as records [key, bound, len, hashes...] into records[k][stride]"
        self.rustcall(self.lib.smgpu_gather_topk_export_raw, self._ptr, self.b._p(records), int(k), records.shape[1], self.b._s())

    def load_candidates(self, records, n):
        "adopt the gathered records of every shard (borrowed until the next load)"
        self._cands = records
        self.rustcall(self.lib.smgpu_gather_cands_load_raw, self._ptr, self.b._p(records), int(n), records.shape[1], self.b._s())

    def replay(self, rounds):
        self.rustcall(self.lib.smgpu_gather_replay_raw, self._ptr, int(rounds), self.b._s())

    def poll(self):
        done = C.c_bool(False)
        rounds = self.rustcall(self.lib.smgpu_gather_poll, self._ptr, C.byref(done), self.b._s())
        return int(rounds), bool(done.value)

    def _fetch(self, fn):
        import numpy as np
        idx, isect = np.zeros(self._cap, dtype=np.uint64), np.zeros(self._cap, dtype=np.uint64)
        n = self.rustcall(fn, self._ptr, idx.ctypes.data_as(C.c_void_p), isect.ctypes.data_as(C.c_void_p), self._cap,
                          self.b._s())
        # tolist(): plain ints in two C loops.  (Building the pairs from numpy scalars allocated enough objects to set
        # off a full cyclic-GC pass over torch's object graph -- 40-70 ms of host time that looked like loop time in
        # tools/bench_gather.py, profiles/r02_gather_host_variance.txt.)
        return list(zip(idx[:n].tolist(), isect[:n].tolist()))

    def results(self):
        return self._fetch(self.lib.smgpu_gather_results)

    def run(self):
        "single GPU: every round enqueued back to back, the host only polls the done flag"
        return self._fetch(self.lib.smgpu_gather_run)

    def loop_eligible(self, n_wg=0):
        "can this index run the resident loop kernel on n_wg workgroups (0: one per CU)?"
        return bool(self.lib.smgpu_gather_loop_eligible(self._ptr, int(n_wg)))

    def launch_shared(self, exchange, rank, run_id, n_wg=0, stream=None):
        """enqueue the armed loop as rank `rank` of the exchange's world (every rank's loop kernel agrees on each round's winner
        through the exchange's host-visible memory); results() waits for it.  -> False if the resident loop does not apply"""
        s = C.c_void_p(stream.cuda_stream) if stream is not None else self.b._s()
        return bool(self.rustcall(self.lib.smgpu_gather_launch_shared, self._ptr, exchange._ptr, int(rank), int(run_id), int(n_wg), s))

    def stats(self):
        """what the index build and the last run() cost (smgpu_gather_stats): kernel spans from HIP events next to the
        host wall clocks, driver-allocator time and calls, host synchronisations -- localises host effects"""
        out = (C.c_double * 9)()
        self.rustcall(self.lib.smgpu_gather_stats, self._ptr, out)
        keys = ("build_kernels_ms", "build_host_ms", "build_driver_alloc_ms", "build_driver_allocs", "build_syncs",
                "build_sync_wait_ms", "loop_gpu_ms", "loop_host_ms", "loop_fallbacks")
        return {k: round(float(v), 3) for k, v in zip(keys, out)}

    def counters(self):
        "remaining overlap of every local dataset (host copy)"
        import numpy as np
        ndb = self._keep[2].numel() - 1
        out = np.zeros(max(ndb, 1), dtype=np.uint64)
        self.rustcall(self.lib.smgpu_gather_counters_get, self._ptr, out.ctypes.data_as(C.c_void_p), self.b._s())
        return out[:ndb]


class GatherExchange:
    """Host-visible memory through which the loop kernels of several ranks agree on every round's winner (smgpu_gather_xchg_*):
    POSIX shared memory registered with HIP when the ranks are processes of one node, private pinned memory when one process
    drives every rank (tests)."""

    transport = "shared host memory"

    def __init__(self, lib, rustcall, world, rowcap, shm_name=None, create=True):
        self.lib, self.world, self.rowcap, self.shm_name = lib, int(world), int(rowcap), shm_name
        self._ptr = rustcall(lib.smgpu_gather_xchg_new, shm_name.encode() if shm_name else None, int(world), int(rowcap), bool(create))

    def __del__(self):
        if getattr(self, "_ptr", None) and self.lib is not None:
            self.lib.smgpu_gather_xchg_free(self._ptr)
            self._ptr = None


class DeviceGatherExchange(GatherExchange):
    """The exchange in DEVICE memory: this rank's area in its own HBM (fine-grained), the peers' areas mapped in through hipIpc
    handles -- between the GPUs of a node the loop kernels then poll each other over xGMI (smgpu_gather_xchg_new_device)."""
    transport = "per-rank device memory mapped through hipIpc (xGMI between the GPUs of a node)"

    def __init__(self, lib, rustcall, world, rank, rowcap):
        self.lib, self.rustcall, self.world, self.rank, self.rowcap, self.shm_name = lib, rustcall, int(world), int(rank), int(rowcap), None
        self._ptr = rustcall(lib.smgpu_gather_xchg_new_device, int(world), int(rank), int(rowcap))

    def export(self):
        buf = (C.c_uint8 * int(self.lib.smgpu_gather_xchg_ipc_handle_size()))()
        self.rustcall(self.lib.smgpu_gather_xchg_ipc_export, self._ptr, buf)
        return bytes(buf)

    def open_peer(self, peer_rank, handle):
        buf = (C.c_uint8 * len(handle)).from_buffer_copy(handle)
        self.rustcall(self.lib.smgpu_gather_xchg_ipc_open, self._ptr, int(peer_rank), buf)


def open_device_exchange(owner, lib, rustcall, world, rank, rowcap, group=None):
    """The device-memory exchange of this job's ranks, made once and reused while it is large enough: every rank allocates its
    area, the IPC handles travel by one all-gather, every rank maps its peers' areas.  Like open_shared_exchange every step is
    agreed before anybody goes on (all ranks end with the exchange, or all raise).  -> (exchange, run id)."""
    dist = _dist()
    cur = getattr(owner, "_xchg_dev", None)
    grouped = world > 1 or (dist.is_available() and dist.is_initialized())
    if cur is None or cur.world != world or cur.rowcap < rowcap or getattr(owner, "_xchg_dev_runs", 0) >= EXCHANGE_RUNS_MAX:
        owner._xchg_dev = None
        made, err = None, None
        try:
            made = DeviceGatherExchange(lib, rustcall, world, rank, rowcap)
        except Exception as e:                                           # noqa: BLE001 -- the peers must learn of it
            err = e
        if grouped and not agree(err is None, group):
            raise err or RuntimeError("the device-memory gather exchange could not be allocated on some rank")
        if err is not None:
            raise err
        if grouped and world > 1:
            handles = [None] * world
            dist.all_gather_object(handles, made.export(), group=group)
            try:
                for r, h in enumerate(handles):
                    if r != rank:
                        made.open_peer(r, h)
            except Exception as e:                                       # noqa: BLE001
                err = e
            if not agree(err is None, group):
                raise err or RuntimeError("a peer's device-memory exchange area could not be mapped on some rank")
        owner._xchg_dev = made
        owner._xchg_dev_runs = 0
    owner._xchg_dev_runs += 1
    return owner._xchg_dev, owner._xchg_dev_runs


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
        """counts for the owned 16-row tiles, entries on or above the diagonal (the driver mirrors them).  method: "merge"
        (LDS hash-table tiles), "bits" (bit rows + popcount) or "auto" (bits when the collection is dense enough)."""
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
            index.compare_tiles(first, stride, count, out=out, upper=True)
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
    def open_exchange(self, world, rank, rowcap, group=None):
        """the exchange through which this job's loop kernels agree on every round (one node).  SMG_GATHER_EXCHANGE:
        `device` -- per-rank device memory mapped through hipIpc (xGMI between GPUs); `shared` -- POSIX shared host memory every
        rank maps and registers with HIP; unset -- the device form, the host form where it cannot be set up (the ranks agree)."""
        import os
        kind = os.environ.get("SMG_GATHER_EXCHANGE", "auto")
        # (a device form that could not be set up once is not tried again by this backend: every attempt costs an allocation, an
        #  IPC export / open and several collectives, and the outcome is agreed among the ranks, so all of them remember alike)
        if kind in ("auto", "device") and world <= 16 and not (kind == "auto" and getattr(self, "_xchg_dev_failed", None)):
            try:
                return open_device_exchange(self, self.lib, self.rustcall, world, rank, rowcap, group)
            except Exception as e:                                           # noqa: BLE001
                if kind == "device":
                    raise
                self._xchg_dev_failed = repr(e)
                import sys
                print(f"[sourmash_amd] rank {rank}: device-memory gather exchange unavailable ({e!r}); using shared host memory",
                      file=sys.stderr)

        def make(name, create):
            return GatherExchange(self.lib, self.rustcall, world, rowcap, name, create=create)
        return open_shared_exchange(self, make, world, rank, rowcap, group)

    def gather_state(self, query, nq, hashes, offsets, ndb, index_base):
        "Invert the shard against the query; -> step object (pick / export / apply / poll / results / run)."
        return _DeviceGatherState(self, query, nq, hashes, offsets, ndb, index_base)

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
def compare_all_pairs_distributed(hashes, offsets, n, backend, group=None, want_jaccard=True, force_collectives=False,
                                  timing=None, result_on="all"):
    """N x N common-hash matrix (int32 bit patterns of u32) and f64 Jaccard on every rank.
    hashes / offsets: the full CSR, replicated on every rank.
    timing (dict, optional; CUDA tensors only): receives tiles_ms / allgather_ms / finish_ms from stream events.
    result_on: "all" -- every rank ends up with both matrices (one all-gather of the counts); "root" -- only rank 0 does, as
    the caller of the reference's compare does (compare.py:14-64 returns ONE matrix): the shards travel to rank 0 alone (a
    gather: each over its own link, an eighth of the all-gather's bytes on the fabric) and the mirror + Jaccard passes run
    there only; the other ranks return (None, None)."""
    if result_on not in ("all", "root"):
        raise ValueError("result_on must be 'all' or 'root'")
    dist = _dist()
    rank, world = world_info(group)
    marks = []

    def mark():
        if timing is not None and hashes.is_cuda:
            ev = backend.torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)

    mark()
    first, stride, count = tiles_for_rank(n, world, rank)
    n_tiles = (n + TILE - 1) // TILE
    if world == 1 and not force_collectives:
        local = backend.compare_tiles(hashes, offsets, n, first, stride, count)
        mark()
        full = local[:n]
    else:
        torch = __import__("torch")
        max_count = (n_tiles + world - 1) // world                # tile slots per rank (the last ranks may own one less)
        # a count is at most the size of the smaller sketch: when no sketch holds 65,536 hashes the shards travel as 16-bit
        # words -- half the bytes on the ring, which is bound per xGMI link (200 MB instead of 400 MB at N = 10,000)
        narrow = bool(n > 0 and int((offsets[1:] - offsets[:-1]).max().item()) < 65536)
        # The exchange is cut into BANDS of tile slots: the all-gather of a band travels (RCCL's own stream) while the next
        # band's tiles are computed -- on 8 ranks the 200 MB exchange, not the tiles, would otherwise bound C4.
        # Still ONE logical exchange of every count, in a few pieces; small problems and host-staged groups keep one piece.
        bands = COMPARE_BANDS if (max_count >= COMPARE_BAND_MIN_SLOTS * COMPARE_BANDS and not _staged(hashes, group)) else 1
        per = (max_count + bands - 1) // bands
        full = backend.empty((n_tiles * TILE, n), torch.int32) if (result_on == "all" or rank == 0) else None
        view = full.view(n_tiles, TILE, n) if full is not None else None
        pieces_sent = 0

        def undeal(s0, s1, pieces, work, _send):
            # wait for the band's collective (orders the current stream behind it), then scatter its pieces into `full`
            if work is not None:
                work.wait()
            if pieces is None:
                return                                             # not the root: its shard is sent, nothing to assemble
            if narrow:
                pieces = [p.view(torch.int16).to(torch.int32).bitwise_and_(0xFFFF) for p in pieces]
            for r in range(world):                                 # un-deal: slot s of rank r is tile r + world * s
                cnt = max(0, min((n_tiles - r + world - 1) // world if n_tiles > r else 0, s1) - s0)
                if cnt:
                    view[r + world * s0: r + world * (s0 + cnt): world] = pieces[r].view(-1, TILE, n)[:cnt]

        # Band b - 1 is un-dealt right after band b's tiles and collective are enqueued: its exchange has had the whole of band
        # b's tiles to travel, and only ONE band's receive pieces and send buffer are alive next to `full` (keeping every band's
        # until the end held about 2 x N x N x 4 bytes per rank, 1.5 x when narrow -- ADVICE r04).
        pending = None
        for b in range(bands):
            s0, s1 = b * per, min(max_count, (b + 1) * per)
            if s1 <= s0:
                break
            mine = min(count, s1) - min(count, s0)                 # this rank's tiles of the band
            local = backend.compare_tiles(hashes, offsets, n, first + s0 * stride, stride, mine)
            if local.shape[0] != (s1 - s0) * TILE:                 # equal-sized pieces for all_gather
                pad = backend.zeros(((s1 - s0) * TILE, n), local.dtype)
                pad[:local.shape[0]] = local
                local = pad
            # (as bytes: neither RCCL nor gloo moves 16-bit integers; truncation keeps the low 16 bits)
            send = local.to(torch.int16).view(torch.uint8) if narrow else local
            if result_on == "root":
                pieces = [backend.empty(tuple(send.shape), send.dtype) for _ in range(world)] if rank == 0 else None
                work = _gather_to_root(pieces, send, group, async_op=bands > 1)
            else:
                pieces = [backend.empty(tuple(send.shape), send.dtype) for _ in range(world)]
                work = _all_gather(pieces, send, group, async_op=bands > 1)
            pieces_sent += 1
            if pending is not None:
                undeal(*pending)
            pending = (s0, s1, pieces, work, send)
        mark()
        if pending is not None:
            undeal(*pending)
        pending = None
        if full is not None:
            full = full[:n].contiguous() if full.shape[0] != n else full
        if timing is not None:
            timing["exchange_bytes_per_entry"] = 2 if narrow else 4
            timing["exchange_pieces"] = pieces_sent
    mark()
    if result_on == "root" and rank != 0:
        full, jac = None, None
    else:
        backend.symmetrize(full, n)
        jac = backend.jaccard(full, offsets, n) if want_jaccard else None
    mark()
    if marks:
        backend.torch.cuda.synchronize()
        timing.update(tiles_ms=marks[0].elapsed_time(marks[1]), allgather_ms=marks[1].elapsed_time(marks[2]),
                      finish_ms=marks[2].elapsed_time(marks[3]))
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


def exchange_geometry(world):
    "-> (K records per rank, replay rounds per exchange)"
    k = max(1, min(TOPK_MAX, CAND_MAX // max(world, 1)))
    # one shard: exactly its K best can win before the kept-back key interferes; W shards: the global order interleaves
    # the shards' lists, and the batch ends when the best shard's K are used up -- about K + (W - 1) K / 2 rounds
    rounds = min(k + (world - 1) * k // 2, world * k)
    return k, max(rounds, 1)


def gather_distributed(query, nq, shard_hashes, shard_offsets, n_shard, index_base, threshold_bp, scaled, backend,
                       group=None, max_rounds=None, stepwise=False, force_collectives=False, stats=None):
    """Min-set-cover gather over a dataset-sharded database.

    query: sorted u64 hashes (int64 bit patterns) replicated on every rank; shard_*: this rank's CSR of
    n_shard sketches whose global indices start at index_base.  Returns [(global index, |intersect|)],
    identical on every rank and identical to the single-process result (ties -> lowest global index).

    Every rank inverts its shard against the query once (backend.gather_state).  Then, per exchange:
        export   the K best local rows as records [key, bound, len, hashes...]      -> ONE all-gather
        load     every rank adopts all W x K records (their counters stay exact from here on)
        replay   R rounds: best candidate (valid while >= every kept-back key; stop rules), apply
    all of it enqueued without a host round trip; the host polls a done flag every few exchanges.  One rank
    without `stepwise` runs the fused native loop (the same kernels, the exchange being a no-op).
    stats (dict, optional): receives exchanges / rounds_per_exchange / records_per_rank / record_words."""
    dist = _dist()
    rank, world = world_info(group)
    torch = backend.torch if hasattr(backend, "torch") else __import__("torch")
    state = backend.gather_state(query, nq, shard_hashes, shard_offsets, n_shard, index_base)
    # search.py:15-37: the float threshold threshold_bp / scaled, compared with integer counts -> its ceiling
    thr = math.ceil(float(threshold_bp) / scaled) if threshold_bp else 0
    collect = world > 1 or force_collectives
    layout_max = backend.zeros((1,), torch.int64)
    layout_sum = backend.zeros((1,), torch.int64)
    layout_max[0], layout_sum[0] = state.longest_row(), n_shard
    if collect:                                            # one-off: longest row anywhere, number of datasets
        _all_reduce(layout_max, dist.ReduceOp.MAX, group)
        _all_reduce(layout_sum, dist.ReduceOp.SUM, group)
    total = int(layout_sum.item())
    stride = CAND_HEAD + max(int(layout_max.item()), 1)
    state.begin(thr, min(max_rounds, total) if max_rounds is not None else total)
    if world == 1 and not stepwise and not force_collectives:
        res = state.run()
        if stats is not None and hasattr(state, "stats"):
            stats.update(state.stats())
        return res
    # Ranks of one node whose indexes can run the resident loop: every rank's loop kernel runs all rounds, the local winners
    # meet in shared host memory each round (csrc/gather.hip: gather_launch_loop) -- no host collective inside the loop.
    # SMG_GATHER_EXCHANGE=records keeps the candidate-record protocol below (what ranks on different nodes would need).
    import os
    # (stepwise=True asks for the record protocol explicitly: tests, and the single-rank comparison of the two)
    if (collect and not stepwise and hasattr(backend, "open_exchange") and hasattr(state, "launch_shared")
            and os.environ.get("SMG_GATHER_EXCHANGE", "auto") != "records"):
        # workgroups per rank's loop kernel: one per CU (0) on a GPU of its own; ranks that SHARE a GPU (tests) must fit side by side
        n_wg = int(os.environ.get("SMG_GATHER_LOOP_WGS", "0"))
        grouped = collect and dist.is_available() and dist.is_initialized()

        def all_ok(flag):
            # Every step is agreed among the ranks before anybody acts on it: a rank that fell back to the record protocol on
            # its own would sit in a collective the others never join.
            return agree(flag, group) if grouped else bool(flag)
        if all_ok(state.loop_eligible(n_wg)):
            res, good, xchg, run_id = None, False, None, 0
            try:
                xchg, run_id = backend.open_exchange(world, rank, stride - CAND_HEAD, group)   # (agreed inside: all ranks or none)
                opened = True
            except Exception:                                   # (shared memory / registration not available here)
                opened = False
            if all_ok(opened):
                try:
                    if state.launch_shared(xchg, rank, run_id, n_wg):
                        res = state.results()                   # waits for the loop; raises if a peer never showed up, or if some
                        good = True                             # rank's grid did not become resident (nothing was applied then)
                except Exception:
                    good = False
                if all_ok(good):
                    if stats is not None:
                        stats.update(exchanges=0, rounds_per_exchange=None, records_per_rank=None, record_words=None, rounds=len(res),
                                     protocol="resident loop kernels, winners agreed every round through " + getattr(xchg, "transport", "shared host memory"))
                        stats.update(state.stats())
                    return res
            # the shared exchange did not work out on some rank: start over with a fresh index and the record protocol
            state = backend.gather_state(query, nq, shard_hashes, shard_offsets, n_shard, index_base)
            state.begin(thr, min(max_rounds, total) if max_rounds is not None else total)
            if stats is not None:
                stats["shared_exchange"] = "failed on some rank; record protocol used"
    k, rounds = exchange_geometry(world)
    mine = backend.zeros((k, stride), torch.int64)
    everyone = backend.zeros((world * k, stride), torch.int64) if collect else mine
    batch, exchanges, done_rounds = 2, 0, 0
    while True:
        for _ in range(batch):
            state.export_topk(mine, k)
            if collect:
                _all_gather_rows(dist, everyone, mine, world, group)       # the ONE collective of an exchange
            state.load_candidates(everyone, world * k)
            state.replay(rounds)
        exchanges += batch
        now, done = state.poll()
        if done:
            break
        # rounds an exchange really yields (replicated state -> the same number on every rank): clustered counters end
        # a batch after a few rounds, and replay rounds past that point are launches that do nothing
        per_exchange = (now - done_rounds) / batch
        done_rounds = now
        rounds = max(2, min(world * k, int(per_exchange * 1.5) + 2))
        batch = min(batch * 2, 16)
    if stats is not None:
        stats.update(exchanges=exchanges, rounds_per_exchange=rounds, records_per_rank=k, record_words=stride,
                     rounds=len(state.results()))
        if hasattr(state, "stats"):
            stats.update(state.stats())
    return state.results()


# ---------------------------------------------------------------------------------------------------
# search / prefetch over a dataset-sharded database (index/__init__.py:115-170,241-256,
# src/core/src/index/linear.rs:52-113): every rank runs ONE overlap pass over its shard, one all-gather hands everyone the
# (|query ∩ row|, |row|) pairs of all rows -- 16 bytes per dataset where the dataset itself is ~40 KB -- and the
# scoring / thresholding / ordering is the single-GPU code on the assembled vectors.

def local_overlaps(query, nq, shard_hashes, shard_offsets, n_shard, backend):
    "this rank's part: int64 [n_shard, 2] = (|query ∩ row|, |row|) for its rows, from one overlap pass"
    torch = backend.torch
    pairs = backend.zeros((max(n_shard, 1), 2), torch.int64)
    if n_shard:
        counts = backend.zeros((n_shard,), torch.int64)
        backend.overlaps(query, nq, shard_hashes, shard_offsets, n_shard, counts, 0)
        pairs[:n_shard, 0] = counts
        pairs[:n_shard, 1] = shard_offsets[1:n_shard + 1] - shard_offsets[:n_shard]
    return pairs[:n_shard]


def assemble_overlaps(pieces, shard_rows):
    "pieces[r]: rank r's padded [longest, 2] block; shard_rows[r]: its number of rows -> [total, 2] in global row order"
    torch = __import__("torch")
    return torch.cat([p[:n] for p, n in zip(pieces, shard_rows)]) if pieces else None


def overlaps_distributed(query, nq, shard_hashes, shard_offsets, n_shard, index_base, backend, group=None,
                         force_collectives=False):
    "-> (shared, sizes): numpy u64 vectors over ALL datasets in global index order, identical on every rank"
    return _overlaps_all(query, nq, shard_hashes, shard_offsets, n_shard, index_base, backend, group, force_collectives)[:2]


def _overlaps_all(query, nq, shard_hashes, shard_offsets, n_shard, index_base, backend, group, force_collectives):
    """-> (shared, sizes, first): `first` = the global index of row 0 of the assembled vectors.
    Shards must be contiguous in the global numbering and ordered by rank (rank r holds [index_base, index_base + n_shard)),
    as bench.py and gather_distributed lay them out; that layout is checked."""
    import numpy as np
    dist = _dist()
    rank, world = world_info(group)
    torch = backend.torch
    mine = local_overlaps(query, nq, shard_hashes, shard_offsets, n_shard, backend)
    if world == 1 and not force_collectives:
        both = mine.cpu().numpy().view(np.uint64)
        return both[:, 0].copy(), both[:, 1].copy(), int(index_base)
    layout = backend.zeros((world, 2), torch.int64)
    me = backend.zeros((1, 2), torch.int64)
    me[0, 0], me[0, 1] = int(n_shard), int(index_base)
    _all_gather_rows(dist, layout, me, world, group)                       # who holds what (16 bytes per rank)
    rows = [int(x) for x in layout[:, 0].tolist()]
    bases = [int(x) for x in layout[:, 1].tolist()]
    if any(bases[r] != bases[0] + sum(rows[:r]) for r in range(world)):
        raise ValueError("overlaps_distributed needs contiguous shards in rank order: bases %r, rows %r" % (bases, rows))
    longest = max(max(rows), 1)
    pad = backend.zeros((longest, 2), torch.int64)
    pad[:n_shard] = mine
    everyone = backend.zeros((world * longest, 2), torch.int64)
    _all_gather_rows(dist, everyone, pad, world, group)                     # the ONE data collective
    full = assemble_overlaps([everyone[r * longest:(r + 1) * longest] for r in range(world)], rows)
    both = full.cpu().numpy().view(np.uint64)
    return both[:, 0].copy(), both[:, 1].copy(), bases[0]


def prefetch_distributed(query, nq, shard_hashes, shard_offsets, n_shard, index_base, threshold_bp, scaled, backend,
                         group=None, force_collectives=False):
    "[(global index, |intersect|)] in index order: datasets sharing >= threshold_bp with the query (search.py:956-976)"
    from .index import prefetch_rows
    shared, _, first = _overlaps_all(query, nq, shard_hashes, shard_offsets, n_shard, index_base, backend, group, force_collectives)
    return [(first + r, c) for r, c in prefetch_rows(shared, threshold_bp, scaled)]


def search_distributed(query, nq, shard_hashes, shard_offsets, n_shard, index_base, backend, group=None, threshold=0.0,
                       do_containment=False, do_max_containment=False, best_only=False, force_collectives=False):
    "[(score, global index)] best first (ties: lowest index) -- the scores of JaccardSearch (search.py:88-160)"
    from .index import rank_search_hits
    shared, sizes, first = _overlaps_all(query, nq, shard_hashes, shard_offsets, n_shard, index_base, backend, group,
                                         force_collectives)
    hits = rank_search_hits(shared, sizes, nq, threshold=threshold, do_containment=do_containment,
                            do_max_containment=do_max_containment, best_only=best_only)
    return [(s, first + r) for s, r in hits]


def gather_emulated_ranks(query, nq, shards, threshold_bp, scaled, backend, max_rounds=None):
    """Diagnostic / test driver: the shared-exchange gather with every rank driven by THIS process on one GPU -- one loop kernel
    per rank, each on its own stream with an equal share of the CUs, a private pinned exchange.  shards: [(hashes, offsets, n,
    index_base)] in rank order.  -> the picks as every rank reports them (a list per rank; they must be identical)."""
    torch = backend.torch
    world = len(shards)
    states = [backend.gather_state(query, nq, h, off, n, base) for h, off, n, base in shards]
    total = sum(n for _, _, n, _ in shards)
    rowcap = max(max(st.longest_row() for st in states), 1)
    thr = math.ceil(float(threshold_bp) / scaled) if threshold_bp else 0
    n_cu = torch.cuda.get_device_properties(backend.device).multi_processor_count
    # a loop workgroup fills a CU and workgroup b lands on XCD b % 8: every rank gets a multiple of 8, and the ranks together leave
    # CUs free (a grid that does not fit is not an error to the runtime -- the surplus workgroup just never starts)
    per = max(8, min(64, (n_cu // world) // 8 * 8))
    if not all(st.loop_eligible(per) for st in states):
        return None
    xchg = GatherExchange(backend.lib, backend.rustcall, world, rowcap)
    backend._emu_runs = getattr(backend, "_emu_runs", 0) + 1
    for st in states:
        st.begin(thr, min(max_rounds, total) if max_rounds is not None else total)
        backend.rustcall(backend.lib.smgpu_gather_loop_reserve, st._ptr, per, rowcap, backend._s())   # no allocation between the launches
    torch.cuda.synchronize()
    # one stream per rank, created once and kept: HIP multiplexes streams onto a few hardware queues, and two loops that land on
    # the same queue run one after the other -- the first would wait for a peer that cannot start (seen with fresh streams
    # on every call: the second call's pair shared a queue)
    pool = backend.__dict__.setdefault("_emu_streams", [])
    while len(pool) < world:
        st_new = torch.cuda.Stream(device=backend.device)
        with torch.cuda.stream(st_new):                      # first use creates the stream's hardware queue (~5 ms): not between launches
            backend.zeros((1,), torch.int64).add_(1)
        pool.append(st_new)
    torch.cuda.synchronize()
    streams = pool[:world]
    for r, (st, stream) in enumerate(zip(states, streams)):
        if not st.launch_shared(xchg, r, backend._emu_runs, per, stream=stream):
            raise RuntimeError("the resident loop did not start for emulated rank %d" % r)
    out = []
    for st, stream in zip(states, streams):
        with torch.cuda.stream(stream):
            out.append(st.results())
    return out


def _all_gather_rows(dist, out, mine, world, group):
    "out[r * k : (r + 1) * k] = rank r's `mine`"
    if _staged(out, group):                                  # a gloo group over device tensors: through the host (see _staged)
        k = mine.shape[0]
        c = mine.cpu().contiguous()
        parts = [c.new_empty(c.shape) for _ in range(world)]
        dist.all_gather(parts, c, group=group)
        for r, part in enumerate(parts):
            out[r * k:(r + 1) * k].copy_(part)
    elif hasattr(dist, "all_gather_into_tensor") and out.is_cuda:
        dist.all_gather_into_tensor(out, mine, group=group)
    else:
        k = mine.shape[0]
        dist.all_gather([out[r * k:(r + 1) * k] for r in range(world)], mine, group=group)
