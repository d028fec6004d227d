"""Device-resident batch API: torch tensors in, torch tensors out.

PyTorch is plumbing here -- it owns the HBM allocations, the current HIP stream
and (in parallel.py) the RCCL process group; every computation is one of the
raw C-ABI entry points of include/sourmash_amd.h (smgpu_*_raw), i.e. the
hand-written HIP kernels.  Nothing in this module runs on the CPU: without a GPU
every function raises.
"""
import ctypes as C

from ._lowlevel import lib
from .minhash import _get_max_hash_for_scaled
from .exceptions import SourmashError, exceptions_by_code
from .utils import decode_str, rustcall


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("sourmash_amd.device needs a HIP device (no CPU fallback)")
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _u64(torch, n, device):
    # torch has no uint64 arithmetic but int64 storage is all we need (bit patterns)
    return torch.empty(int(n), dtype=torch.int64, device=device)


def arena_stats():
    "the library's device arena (csrc/arena.hpp): driver calls, time in the driver, reuse hits, bytes live / cached"
    out = (C.c_uint64 * 8)()
    lib.smgpu_arena_stats(out)
    keys = ("driver_allocs", "driver_frees", "driver_ns", "reuse_hits", "live_bytes", "cached_bytes", "peak_bytes",
            "cross_stream_waits")
    return dict(zip(keys, (int(v) for v in out)))


def arena_trim(keep_bytes=0):
    "give the arena's cached blocks back to the driver"
    lib.smgpu_arena_trim(int(keep_bytes))


def synth_dna(n, seed=42, record_len=0, start=0, device="cuda", out=None):
    "n bytes of the BASELINE C2 random-DNA stream generated in HBM (uint8 tensor)."
    torch = _torch()
    if out is None:
        out = torch.empty(int(n), dtype=torch.uint8, device=device)
    rustcall(lib.smgpu_synth_dna_raw, _ptr(out), int(start), int(n), int(seed), int(record_len), _stream(torch))
    return out


class DeviceSketcher:
    """Reusable scratch for sketching device-resident sequence buffers.

    sketch(seq) -> int64 tensor viewing the sorted unique kept hashes (u64 bit patterns).
    Capacity is sized from scaled (expected kept = len/scaled) with slack; on
    overflow the buffers grow and the call is repeated.
    """

    def __init__(self, ksize=31, scaled=1000, seed=42, device="cuda"):
        self.torch = _torch()
        self.ksize, self.scaled, self.seed, self.device = int(ksize), int(scaled), int(seed), device
        self.max_hash = _get_max_hash_for_scaled(scaled)
        self.cap = 0
        self.out = self.ws = None
        self.result = self.torch.zeros(2, dtype=self.torch.int64, device=device)

    def _reserve(self, cap):
        if cap <= self.cap:
            return
        torch = self.torch
        self.cap = int(cap)
        self.out = _u64(torch, self.cap, self.device)
        nbytes = lib.smgpu_sketch_workspace_bytes(self.cap)
        self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)

    def capacity_for(self, n_bases):
        expect = n_bases / max(self.scaled, 1)
        return int(expect * 1.25 + 8 * expect ** 0.5 + 4096)

    def sketch(self, seq):
        torch = self.torch
        assert seq.dtype == torch.uint8 and seq.is_cuda and seq.is_contiguous()
        n = seq.numel()
        self._reserve(self.capacity_for(n))
        for attempt in range(2):
            lib.sourmash_err_clear()
            got = lib.smgpu_sketch_dna_raw(_ptr(seq), n, self.ksize, self.seed, self.max_hash, _ptr(self.out),
                                           self.cap, _ptr(self.result), _ptr(self.ws), self.ws.numel(),
                                           _stream(torch))
            code = lib.sourmash_err_get_last_code()
            if code == 0:
                return self.out[:got]
            message = decode_str(lib.sourmash_err_get_last_message())
            kept = int(self.result[0].item())
            if attempt == 0 and kept > self.cap:   # repetitive input beat the estimate: grow and retry once
                self._reserve(kept + 1024)
                continue
            raise exceptions_by_code.get(code, SourmashError)(message)

    def kernel_only(self, seq, out, count):
        "Just the k-mer kernel (no sort): appends to `out`, adds to `count` (int64[1], caller zeroes)."
        torch = self.torch
        rustcall(lib.smgpu_sketch_dna_kernel_raw, _ptr(seq), seq.numel(), self.ksize, self.seed, self.max_hash,
                 _ptr(out), out.numel(), _ptr(count), _stream(torch))


def sort_unique(keys):
    """int64 tensor of u64 bit patterns, any order, duplicates allowed -> the sorted (as unsigned) distinct values, a new
    tensor: the library's radix sort + run-length encode (smgpu_sort_unique_raw, csrc/device_sort.hip).  `keys` is consumed."""
    torch = _torch()
    assert keys.dtype == torch.int64 and keys.is_cuda and keys.is_contiguous()
    n = keys.numel()
    out = _u64(torch, max(n, 1), keys.device)
    if n == 0:
        return out[:0]
    ws = torch.empty(int(lib.smgpu_sketch_workspace_bytes(n)), dtype=torch.uint8, device=keys.device)
    n_out = torch.zeros(1, dtype=torch.int64, device=keys.device)
    m = rustcall(lib.smgpu_sort_unique_raw, _ptr(keys), n, _ptr(out), _ptr(n_out), _ptr(ws), ws.numel(), _stream(torch))
    return out[:m]


def pack_csr(sketches, device="cuda"):
    "list of sorted u64 numpy arrays -> (hashes int64 tensor, offsets int64 tensor) on device."
    import numpy as np
    torch = _torch()
    offsets = np.zeros(len(sketches) + 1, dtype=np.int64)
    for i, s in enumerate(sketches):
        offsets[i + 1] = offsets[i] + len(s)
    flat = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches]) if len(sketches) and offsets[-1] \
        else np.zeros(0, dtype=np.uint64)
    hashes = torch.from_numpy(flat.view(np.int64).copy()).to(device)
    if hashes.numel() == 0:
        hashes = torch.zeros(2, dtype=torch.int64, device=device)
    return hashes, torch.from_numpy(offsets).to(device)


class BitIndex:
    """Compare index of a device CSR (smgpu_bitindex_*): hashes held by many sketches as bit columns, hashes held
    by few as inverted lists.  `BitIndex.build` returns None when the merge kernel is the cheaper tool."""

    def __init__(self, ptr, n):
        self._ptr, self.n = ptr, n

    @classmethod
    def build(cls, hashes, offsets, threshold=None, one_shot=False):
        """threshold: hashes held by more sketches than this become bit columns (None: the library's cost model);
        one_shot: the index serves one compare only, so its own build time counts against it."""
        torch = _torch()
        n = offsets.numel() - 1
        ptr = rustcall(lib.smgpu_bitindex_new_ex, _ptr(hashes), _ptr(offsets), n, 0, int(threshold or 0), bool(one_shot),
                       _stream(torch))
        return cls(ptr, n) if ptr else None

    @property
    def universe(self):
        return lib.smgpu_bitindex_universe(self._ptr)

    @property
    def builder(self):
        "which builder made the index: 'dictionary' (sort-free passes, csrc/dictindex.hip) or 'sort' (radix sort of all pairs)"
        return {1: "dictionary", 2: "sort"}.get(int(lib.smgpu_bitindex_builder(self._ptr)), "?")

    @property
    def stats(self):
        "(frequent hashes = bit columns, matrix increments the rare hashes cost per compare, threshold)"
        f, r, t = C.c_uint64(), C.c_uint64(), C.c_uint32()
        lib.smgpu_bitindex_stats(self._ptr, C.byref(f), C.byref(r), C.byref(t))
        return f.value, r.value, t.value

    def compare_tiles(self, first, stride, count, out=None, upper=False):
        """u32 counts for the 16-row tiles first, first+stride, ... (count of them): all columns, or with upper=True only
        the entries on or above the diagonal (for callers that mirror the triangle afterwards; the rest is left as it was)"""
        torch = _torch()
        if out is None:
            out = torch.empty((count * 16, self.n), dtype=torch.int32, device="cuda")
        fn = lib.smgpu_bitindex_compare_upper_raw if upper else lib.smgpu_bitindex_compare_raw
        rustcall(fn, self._ptr, first, stride, count, _ptr(out), _stream(torch))
        return out

    def __del__(self):
        if getattr(self, "_ptr", None) and lib is not None:        # (module globals are gone at interpreter exit)
            lib.smgpu_bitindex_free(self._ptr)
            self._ptr = None


def compare_rows(hashes, offsets, row_lo=0, row_hi=None, want_jaccard=True, common=None, jaccard=None, index=None,
                 method="merge"):
    """common[(row_hi-row_lo), n] (int32 view of u32) and jaccard (float64) for a row block of the
    all-pairs matrix of a device CSR.  Asynchronous on the current stream.

    index: a BitIndex built for this CSR -> indexed path (bit columns + inverted lists).  Without one, method
    "merge" runs the LDS-tiled merge kernel and "auto" first lets the library build a one-shot index if its cost
    model says that beats merging (what smgpu_compare_all_pairs does)."""
    torch = _torch()
    if index is None and method == "auto":
        index = BitIndex.build(hashes, offsets, one_shot=True)
    n = offsets.numel() - 1
    row_hi = n if row_hi is None else row_hi
    rows = row_hi - row_lo
    if want_jaccard and jaccard is None:
        jaccard = torch.empty((rows, n), dtype=torch.float64, device=hashes.device)
    if index is not None and row_lo % 16 == 0:
        count = (rows + 15) // 16
        if common is None or common.shape[0] < count * 16:
            common = torch.empty((count * 16, n), dtype=torch.int32, device=hashes.device)
        if row_lo == 0 and row_hi == n:                         # the whole matrix: the triangle, then its mirror image
            index.compare_tiles(0, 1, count, out=common, upper=True)
            rustcall(lib.smgpu_symmetrize_raw, _ptr(common), n, _stream(torch))
        else:
            index.compare_tiles(row_lo // 16, 1, count, out=common)
        if want_jaccard:
            rustcall(lib.smgpu_jaccard_raw, _ptr(common), _ptr(offsets), n, row_lo, row_hi, _ptr(jaccard), _stream(torch))
        return common[:rows], (jaccard if want_jaccard else None)
    if common is None:
        common = torch.empty((rows, n), dtype=torch.int32, device=hashes.device)
    rustcall(lib.smgpu_compare_raw, _ptr(hashes), _ptr(offsets), n, row_lo, row_hi, _ptr(common),
             _ptr(jaccard) if want_jaccard else None, _stream(torch))
    return common, (jaccard if want_jaccard else None)
