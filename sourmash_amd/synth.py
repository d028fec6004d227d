"""Synthetic sketch collections for the compare / gather configurations
(BASELINE.json configs C3-C5; generators of SURVEY.md section 8d).  Pure numpy,
deterministic, identical on every host -- inputs only, no product logic.
"""
import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
MAX_HASH_1000 = 18446744073709552


def splitmix64(x):
    "vectorised splitmix64 finaliser over uint64 arrays (wrapping arithmetic)."
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def synth_sketches(n, seed=1234, pool_size=50_000, keep_one_in=10, max_hash=MAX_HASH_1000, planted=True):
    """n sorted unique u64 sketches of ~pool_size/keep_one_in hashes drawn from a shared pool
    (E|A ∩ B| ~ pool_size/keep_one_in^2), plus planted edge rows at the end when planted:
    a duplicate of row 0, a sketch disjoint from the pool, a 1-hash sketch, the whole pool."""
    j = np.arange(pool_size, dtype=np.uint64)
    pool = np.unique(splitmix64(np.uint64(seed) + j) % np.uint64(max_hash + 1))
    pool = pool[pool > 0]
    n_plain = n - 4 if planted and n >= 8 else n
    out = []
    jj = np.arange(len(pool), dtype=np.uint64)
    for i in range(n_plain):
        key = (np.uint64(i) << np.uint64(32)) ^ jj ^ np.uint64(0x9E3779B97F4A7C15) ^ np.uint64(seed)
        sel = splitmix64(key) % np.uint64(keep_one_in) == 0
        out.append(pool[sel])
    if n_plain != n:
        out.append(out[0].copy())                                        # identical pair -> jaccard 1.0
        priv = np.unique(splitmix64((np.uint64(1) << np.uint64(61)) + np.uint64(seed) + np.arange(5000, dtype=np.uint64)) % np.uint64(max_hash + 1))
        out.append(np.setdiff1d(priv[priv > 0], pool))                   # no overlap with anything
        out.append(pool[:1].copy())                                      # 1-hash sketch
        out.append(pool.copy())                                          # 50k-hash sketch (superset of all)
    return out


def synth_gather(n_query=1_000_000, n_db=100_000, db_size=5000, seed=777, max_hash=MAX_HASH_1000):
    """(query, [db sketches]): query = n_query distinct hashes; every db sketch draws half of its
    ~db_size hashes from the query and half from a private stream (config C5)."""
    q = np.unique(splitmix64(np.uint64(seed) + np.arange(int(n_query * 1.01), dtype=np.uint64)) % np.uint64(max_hash + 1))
    q = q[q > 0][:n_query]
    half = db_size // 2
    db = []
    for d in range(n_db):
        idx = splitmix64((np.uint64(d) << np.uint64(32)) ^ np.arange(half, dtype=np.uint64) ^ np.uint64(seed)) % np.uint64(len(q))
        shared = q[np.unique(idx)]
        priv = splitmix64((np.uint64(1) << np.uint64(62)) + (np.uint64(d) << np.uint64(33)) + np.arange(half, dtype=np.uint64) + np.uint64(seed)) % np.uint64(max_hash + 1)
        db.append(np.unique(np.concatenate([shared, priv[priv > 0]])))
    return q, db


# ---- device-side generator of config C5 (torch tensors in HBM; no host copy of the 4 GB database) ----------------
def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def splitmix63(x):
    "splitmix64 finaliser on int64 tensors (wrapping), top bit dropped -> non-negative"
    def c(v):
        return v - (1 << 64) if v >= (1 << 63) else v
    x = x + c(0x9E3779B97F4A7C15)
    x = (x ^ _lsr(x, 30)) * c(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * c(0x94D049BB133111EB)
    x = x ^ _lsr(x, 31)
    return _lsr(x, 1)


def synth_gather_device(nq, ndb, dbsize, dev, seed=777, chunk=10_000, row_lo=0, row_hi=None):
    """(query, hashes, offsets) as int64 tensors on `dev`: the construction of synth_gather with hashes drawn as
    (splitmix64(x) >>> 1) mod m so that torch's signed int64 arithmetic can express it.  query = nq distinct hashes
    below max_hash(scaled=1000); database row d takes half of its hashes from the query and half from a private
    stream that depends on d only -- rows [row_lo, row_hi) of the ndb-row database are generated (a rank's shard)."""
    import torch
    row_hi = ndb if row_hi is None else row_hi
    q = torch.unique(splitmix63(torch.arange(int(nq * 1.01) + 16, device=dev, dtype=torch.int64) + seed) % (MAX_HASH_1000 + 1))
    q = q[q > 0][:nq].contiguous()
    half = dbsize // 2
    rows, lens = [], []
    col = torch.arange(half, device=dev, dtype=torch.int64)
    for lo in range(row_lo, row_hi, chunk):
        d = torch.arange(lo, min(lo + chunk, row_hi), device=dev, dtype=torch.int64)[:, None]
        shared = q[splitmix63((d << 32) ^ col[None, :] ^ seed) % len(q)]
        priv = splitmix63((1 << 62) + (d << 33) + col[None, :] + seed) % MAX_HASH_1000 + 1
        x = torch.sort(torch.cat([shared, priv], dim=1), dim=1).values
        keep = torch.ones_like(x, dtype=torch.bool)
        keep[:, 1:] = x[:, 1:] != x[:, :-1]
        rows.append(x[keep])
        lens.append(keep.sum(dim=1))
    n_rows = row_hi - row_lo
    hashes = torch.cat(rows) if rows else torch.zeros(2, dtype=torch.int64, device=dev)
    offsets = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    if rows:
        offsets[1:] = torch.cumsum(torch.cat(lens), 0)
    return q, hashes, offsets


def synth_sketches_device(n, dev, seed=1234, pool_size=50_000, keep_one_in=10, chunk=2000):
    """(hashes, offsets) int64 tensors on `dev`: n sketches drawn from a shared pool like synth_sketches (every pool hash kept
    with probability 1 / keep_one_in, decided by splitmix of (row, pool index)), generated in HBM -- for collections too large
    to build on the host in a benchmark's time (40,000 sketches = 2e8 hashes).  Identical on every rank."""
    import torch
    pool = torch.unique(splitmix63(torch.arange(pool_size, device=dev, dtype=torch.int64) + seed) % MAX_HASH_1000 + 1)
    jj = torch.arange(pool.numel(), device=dev, dtype=torch.int64)
    rows, lens = [], []
    for lo in range(0, n, chunk):
        i = torch.arange(lo, min(lo + chunk, n), device=dev, dtype=torch.int64)[:, None]
        sel = splitmix63((i << 32) ^ jj[None, :] ^ (seed * 2654435761 % (1 << 31))) % keep_one_in == 0
        rows.append(pool[None, :].expand(sel.shape[0], -1)[sel])      # row-major: every row's hashes in pool (= sorted) order
        lens.append(sel.sum(dim=1))
    hashes = torch.cat(rows)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(torch.cat(lens), 0)
    return hashes, offsets
