"""`sourmash sketch dna` driver: parameter strings, FASTA/FASTQ ingest, sketching.

Thin host layer around the GPU path:
  src/sourmash/command_sketch.py:33-87    parameter-string grammar ("k=31,scaled=1000,abund")
  src/sourmash/command_sketch.py:90-186   one signature template per parameter string
  src/sourmash/command_sketch.py:662-832  per-file driver: every record of a file accumulates
                                          into the same signatures (unless singleton)
  src/sourmash/command_sketch.py:864-1085 ComputeParameters over computeparams_*
Where the reference crosses the FFI once per FASTA record, `sketch_file` joins the
records of a file with a separator byte and crosses it once per file: a byte
outside ACGT kills exactly the k-mers that would span two records, which is what
one add_sequence call per record achieves (force=True semantics, the CLI default).
"""
import argparse
import ctypes as C
import gzip
import io
import sys

from ._lowlevel import ffi, lib
from .minhash import MINHASH_DEFAULT_SEED
from .signature import SourmashSignature
from .utils import RustObject, rustcall

DEFAULTS = dict(dna="k=31,scaled=1000,noabund", protein="k=10,scaled=200,noabund",      # command_sketch.py:25-30
                dayhoff="k=16,scaled=200,noabund", hp="k=42,scaled=200,noabund")
MIN_SCALED, MAX_SCALED = 100, 1e6                 # advisory bounds of sourmash_args.py:61-82 (warnings there, not errors)


def _notify(msg):
    print(msg, file=sys.stderr)


def check_scaled_bounds(arg):
    "negative is an error, outside [100, 1e6] a warning (sourmash_args.py:61-70)"
    f = float(arg)
    if f < 0:
        raise argparse.ArgumentTypeError("ERROR: scaled value must be positive")
    if f < MIN_SCALED:
        _notify("WARNING: scaled value should be >= 100. Continuing anyway.")
    if f > MAX_SCALED:
        _notify("WARNING: scaled value should be <= 1e6. Continuing anyway.")
    return f


def check_num_bounds(arg):
    "negative is an error, outside [50, 50000] a warning (sourmash_args.py:73-82)"
    f = int(arg)
    if f < 0:
        raise argparse.ArgumentTypeError("ERROR: num value must be positive")
    if f < 50:
        _notify("WARNING: num value should be >= 50. Continuing anyway.")
    if f > 50000:
        _notify("WARNING: num value should be <= 50000. Continuing anyway.")
    return f


def parse_params_str(params_str):
    "-> (moltype or None, dict(ksize=[...], num=, scaled=, seed=, track_abundance=)); command_sketch.py:33-87"
    moltype, params = None, {"ksize": []}
    for item in params_str.split(","):
        if item == "abund":
            params["track_abundance"] = True
        elif item == "noabund":
            params["track_abundance"] = False
        elif item in ("protein", "dayhoff", "hp", "dna"):
            moltype = item
        elif item.startswith("scaled"):
            if len(item) < 8 or item[6] != "=":
                raise ValueError("scaled takes a parameter, e.g. 'scaled=1000'")
            if params.get("num"):
                raise ValueError("cannot set both num and scaled in a single minhash")
            try:
                scaled = int(item[7:])
            except ValueError:
                raise ValueError(f"cannot parse scaled='{item[7:]}' as an integer")
            params["scaled"], params["num"] = int(check_scaled_bounds(scaled)), 0
        elif item.startswith("seed"):
            if len(item) < 6 or item[4] != "=":
                raise ValueError("seed takes a parameter, e.g. 'seed=42'")
            params["seed"] = int(item[5:])
        elif item.startswith("num"):
            if len(item) < 5 or item[3] != "=":
                raise ValueError("num takes a parameter, e.g. 'num=500'")
            if params.get("scaled"):
                raise ValueError("cannot set both num and scaled in a single minhash")
            try:
                num = int(item[4:])
            except ValueError:
                raise ValueError(f"cannot parse num='{item[4:]}' as a number")
            params["num"], params["scaled"] = check_num_bounds(num), 0
        elif item.startswith("k"):
            if len(item) < 3 or item[1] != "=":
                raise ValueError("k takes a parameter, e.g. 'k=31'")
            params["ksize"].append(int(item[2:]))
        else:
            raise ValueError(f"unknown component '{item}' in params string")
    return moltype, params


class ComputeParameters(RustObject):
    "Parameter block that crosses the ABI (command_sketch.py:864-1085 over ffi/cmd/compute.rs)."
    __dealloc_func__ = lib.computeparams_free

    def __init__(self, *, ksizes=(21, 31, 51), seed=42, protein=False, dayhoff=False, hp=False, dna=True,
                 num_hashes=500, track_abundance=False, scaled=0):
        self._objptr = lib.computeparams_new()
        self.seed = seed
        self.ksizes = ksizes
        self.protein, self.dayhoff, self.hp, self.dna = protein, dayhoff, hp, dna
        self.num_hashes = num_hashes
        self.track_abundance = track_abundance
        self.scaled = scaled

    @classmethod
    def from_param_str(cls, params_str, default_moltype="dna"):
        "One parameter string layered over the moltype defaults (command_sketch.py:90-186)."
        moltype, params = parse_params_str(params_str)
        moltype = moltype or default_moltype
        _, merged = parse_params_str(DEFAULTS[moltype])
        if params["ksize"]:
            merged["ksize"] = params["ksize"]
        if moltype != "dna":
            merged["ksize"] = [k * 3 for k in merged["ksize"]]      # residues -> stored ksize (command_sketch.py:156)
        for key in ("seed", "track_abundance"):
            if key in params:
                merged[key] = params[key]
        if "num" in params or "scaled" in params:
            merged["num"], merged["scaled"] = params.get("num", 0), params.get("scaled", 0)
        return cls(ksizes=merged["ksize"], seed=merged.get("seed", MINHASH_DEFAULT_SEED), dna=moltype == "dna",
                   protein=moltype == "protein", dayhoff=moltype == "dayhoff", hp=moltype == "hp",
                   num_hashes=merged.get("num", 0), track_abundance=merged.get("track_abundance", False),
                   scaled=merged.get("scaled", 0))

    def _flag(name):   # noqa: N805
        getter, setter = getattr(lib, "computeparams_" + name), getattr(lib, "computeparams_set_" + name)
        return property(lambda self: self._methodcall(getter), lambda self, v: self._methodcall(setter, v))

    seed = _flag("seed")
    protein = _flag("protein")
    dayhoff = _flag("dayhoff")
    hp = _flag("hp")
    dna = _flag("dna")
    num_hashes = _flag("num_hashes")
    track_abundance = _flag("track_abundance")
    scaled = _flag("scaled")
    del _flag

    @property
    def ksizes(self):
        size = ffi.new_size()
        ptr = self._methodcall(lib.computeparams_ksizes, C.byref(size))
        try:
            return [ptr[i] for i in range(size.value)]
        finally:
            lib.computeparams_ksizes_free(ptr, size.value)

    @ksizes.setter
    def ksizes(self, v):
        v = list(v)
        self._methodcall(lib.computeparams_set_ksizes, (C.c_uint32 * max(len(v), 1))(*v), len(v))

    @property
    def moltype(self):
        assert self.dna or self.protein or self.hp or self.dayhoff
        return "DNA" if self.dna else "protein" if self.protein else "hp" if self.hp else "dayhoff"

    @classmethod
    def from_manifest_row(cls, row):
        "the parameters that rebuild the sketch a manifest row describes (command_sketch.py:892-924)"
        moltype = row["moltype"]
        assert moltype in ("DNA", "protein", "hp", "dayhoff")
        return cls(ksizes=[row["ksize"] if moltype == "DNA" else row["ksize"] * 3], seed=MINHASH_DEFAULT_SEED,
                   protein=moltype == "protein", dayhoff=moltype == "dayhoff", hp=moltype == "hp", dna=moltype == "DNA",
                   num_hashes=row["num"], track_abundance=row["with_abundance"], scaled=row["scaled"])

    def to_param_str(self):
        "the parameter string that gives these parameters back: ksizes in residues, defaults left out (:926-964)"
        parts = ["dna" if self.dna else "protein" if self.protein else "hp" if self.hp else "dayhoff"]
        parts += [f"k={k if self.dna else k // 3}" for k in self.ksizes]
        assert self.num_hashes or self.scaled
        parts.append(f"num={self.num_hashes}" if self.num_hashes else f"scaled={self.scaled}")
        if self.track_abundance:
            parts.append("abund")
        if self.seed != MINHASH_DEFAULT_SEED:
            parts.append(f"seed={self.seed}")
        return ",".join(parts)

    def __repr__(self):
        return (f"ComputeParameters(ksizes={self.ksizes}, seed={self.seed}, protein={self.protein}, dayhoff={self.dayhoff}, "
                f"hp={self.hp}, dna={self.dna}, num_hashes={self.num_hashes}, track_abundance={self.track_abundance}, "
                f"scaled={self.scaled})")

    def __eq__(self, other):
        return all(getattr(self, k) == getattr(other, k) for k in
                   ("ksizes", "seed", "protein", "dayhoff", "hp", "dna", "num_hashes", "track_abundance", "scaled"))

    @staticmethod
    def from_args(args):
        "every attribute of an argparse namespace that names a parameter (command_sketch.py:982-993)"
        ret = ComputeParameters._from_objptr(lib.computeparams_new())
        for arg, value in vars(args).items():
            prop = getattr(ComputeParameters, arg, None)
            if isinstance(prop, property) and prop.fset is not None:
                prop.fset(ret, value)
        return ret


class _signatures_for_sketch_factory:                        # noqa: N801  (the reference's name, command_sketch.py:90)
    """Signature templates for a list of parameter strings: each string is laid over the defaults of its molecule
    type; `default_moltype` is the subcommand's (dna / protein / translate input)."""

    def __init__(self, params_str_list, default_moltype):
        self.defaults = {}
        for moltype, pstr in DEFAULTS.items():
            mt, d = parse_params_str(pstr)
            assert mt is None
            self.defaults[moltype] = d
        self.params_list = []
        self.mult_ksize_by_3 = True
        if not params_str_list:
            if default_moltype is None:
                raise ValueError("No default moltype and none specified in param string")
            self.params_list.append((default_moltype, {}))
            return
        for params_str in params_str_list:
            moltype, params = parse_params_str(params_str)
            if moltype and moltype != "dna" and default_moltype == "dna":
                raise ValueError(f"Incompatible sketch type ({default_moltype}) and parameter override ({moltype}) in "
                                 f"'{params_str}'; maybe use 'sketch translate'?")
            if moltype == "dna" and default_moltype and default_moltype != "dna":
                raise ValueError(f"Incompatible sketch type ({default_moltype}) and parameter override ({moltype}) in "
                                 f"'{params_str}'")
            if moltype is None:
                if default_moltype is None:
                    raise ValueError("No default moltype and none specified in param string")
                moltype = default_moltype
            self.params_list.append((moltype, params))

    def get_compute_params(self, *, split_ksizes=False):
        for moltype, given in self.params_list:
            dflt = self.defaults[moltype]
            ksizes = given.get("ksize") or dflt["ksize"]
            if self.mult_ksize_by_3 and moltype != "dna":
                ksizes = [k * 3 for k in ksizes]              # residues -> stored ksize

            def make(ks):
                return ComputeParameters(ksizes=ks, seed=given.get("seed", dflt.get("seed", MINHASH_DEFAULT_SEED)),
                                         protein=moltype == "protein", dayhoff=moltype == "dayhoff", hp=moltype == "hp",
                                         dna=moltype == "dna", num_hashes=given.get("num", dflt.get("num", 0)),
                                         track_abundance=given.get("track_abundance", dflt["track_abundance"]),
                                         scaled=given.get("scaled", dflt.get("scaled", 0)))
            if split_ksizes:
                for k in ksizes:
                    yield make([k])
            else:
                yield make(ksizes)

    def __call__(self, *, split_ksizes=False):
        "fresh signatures, one per parameter set"
        return [SourmashSignature.from_params(p) for p in self.get_compute_params(split_ksizes=split_ksizes)]


# ---- FASTA / FASTQ ingest (screed replacement) --------------------
def _open_maybe_gzip(path):
    fh = open(path, "rb")
    magic = fh.read(2)
    fh.seek(0)
    return gzip.open(fh, "rb") if magic == b"\x1f\x8b" else fh


def read_records(path):
    "Yield (name, sequence-bytes) from FASTA or FASTQ (optionally gzip); newlines stripped."
    with _open_maybe_gzip(path) as fh:
        data = io.BufferedReader(fh) if not isinstance(fh, io.BufferedReader) else fh
        first = data.peek(1)[:1]
        if first == b"@":                                    # FASTQ: 4-line records
            while True:
                head = data.readline()
                if not head:
                    return
                seq = data.readline().rstrip(b"\r\n")
                data.readline()
                data.readline()
                yield head[1:].rstrip(b"\r\n").decode("utf-8", "replace"), seq
        name, chunks = None, []
        for line in data:
            if line.startswith(b">"):
                if name is not None:
                    yield name, b"".join(chunks)
                name, chunks = line[1:].rstrip(b"\r\n").decode("utf-8", "replace"), []
            elif name is not None:
                chunks.append(line.rstrip(b"\r\n"))
        if name is not None:
            yield name, b"".join(chunks)


def sketch_records(records, param_str=DEFAULTS["dna"], *, name="", filename="", check_sequence=False,
                   singleton=False, moltype="dna", input_is_protein=False):
    """Sketch an iterable of (name, sequence) records.

    singleton=False: one SourmashSignature (all ksizes) for the whole input
    (command_sketch.py:741-768); singleton=True: one per record (:712-739).
    check_sequence=True maps to force=False (cli/sketch/dna.py:42-46).
    moltype protein / dayhoff / hp: `sourmash sketch protein` (input_is_protein=True, records are residues) or
    `sourmash sketch translate` (records are DNA, translated in six frames); one call per record."""
    params = ComputeParameters.from_param_str(param_str, default_moltype=moltype)
    force = not check_sequence
    if not params.dna:
        add = (lambda sig, seq: sig.add_protein(seq)) if input_is_protein else (lambda sig, seq: sig.add_sequence(seq, force))
        if singleton:
            out = []
            for rec_name, seq in records:
                sig = SourmashSignature.from_params(params)
                add(sig, seq)
                sig.name = rec_name
                if filename:
                    sig.filename = filename
                out.append(sig)
            return out
        sig = SourmashSignature.from_params(params)
        for _, seq in records:
            add(sig, seq)
        if name:
            sig.name = name
        if filename:
            sig.filename = filename
        return [sig]
    if singleton:
        out = []
        for rec_name, seq in records:
            sig = SourmashSignature.from_params(params)
            sig.add_sequence(seq, force)
            sig.name = rec_name
            if filename:
                sig.filename = filename
            out.append(sig)
        return out
    sig = SourmashSignature.from_params(params)
    if force:
        # one FFI crossing per file: records joined by a byte that no k-mer may span
        joined = b"\n".join(seq for _, seq in records)
        for mh_sig in [sig]:
            _add_buffer_to_signature(mh_sig, joined)
    else:
        for _, seq in records:
            sig.add_sequence(seq, False)
    if name:
        sig.name = name
    if filename:
        sig.filename = filename
    return [sig]


def _add_buffer_to_signature(sig, buf):
    "Feed one buffer to every sketch of a signature through the batch entry point."
    mhs = sig.minhashes()
    if len(mhs) == 1:
        mh = mhs[0].to_mutable()
        mh.add_sequence_buffer(buf, force=True)
        sig.minhash = mh
        return
    # several sketches: rebuild the signature sketch by sketch
    first = True
    for fmh in mhs:
        mh = fmh.to_mutable()
        mh.add_sequence_buffer(buf, force=True)
        if first:
            rustcall(lib.signature_set_mh, sig._get_objptr(), mh._get_objptr())
            first = False
        else:
            rustcall(lib.signature_push_mh, sig._get_objptr(), mh._get_objptr())


def sketch_file(path, param_str=DEFAULTS["dna"], *, name=None, check_sequence=False, singleton=False, moltype="dna",
                input_is_protein=False):
    """`sourmash sketch dna -p <param_str> <path>` -> list of SourmashSignature.

    Default mode (one signature per file, force=True) runs the native streaming ingest
    (smgpu_signature_add_file: C++ FASTA/FASTQ(.gz) reader -> pinned buffers -> GPU, every ksize in one
    pass); --singleton and --check-sequence go record by record like the reference."""
    params = ComputeParameters.from_param_str(param_str, default_moltype=moltype)
    if not singleton and not check_sequence and params.dna:
        sig = SourmashSignature.from_params(params)
        n_records = C.c_uint64(0)
        rustcall(lib.smgpu_signature_add_file, sig._get_objptr(), str(path).encode("utf-8"), C.byref(n_records))
        if name:
            sig.name = name
        sig.filename = str(path)
        return [sig]
    recs = list(read_records(path))
    if name is None and not singleton and recs:
        name = ""
    return sketch_records(recs, param_str, name=name or "", filename=str(path), check_sequence=check_sequence,
                          singleton=singleton, moltype=moltype, input_is_protein=input_is_protein)


def sketch_files(paths, param_str=DEFAULTS["dna"], *, threads=0):
    """`sourmash sketch dna -p <param_str> f1 f2 ...` -> one SourmashSignature per file, in order.

    The files stream through `threads` independent pipelines (host reader / inflate thread -> pinned ring -> HBM ->
    device-side parse -> sketch), so a directory of gzipped genomes is no longer bound by one zlib stream."""
    paths = [str(p) for p in paths]
    if not paths:
        return []
    params = ComputeParameters.from_param_str(param_str)
    arr = (C.c_char_p * len(paths))(*[p.encode("utf-8") for p in paths])
    bases = C.c_uint64(0)
    ptr = rustcall(lib.smgpu_sketch_files, arr, len(paths), params._get_objptr(), int(threads), C.byref(bases))
    sigs = [SourmashSignature._from_objptr(ptr[i]) for i in range(len(paths))]
    lib.nodegraph_buffer_free(C.cast(ptr, C.POINTER(C.c_uint8)), len(paths) * C.sizeof(C.c_void_p))
    return sigs


def gunzip_files(paths, capacity=None):
    """The device inflater behind the .gz ingest by itself (smgpu_gunzip_files): single-member gzip files -> their bytes.

    -> (list of bytes or None per file -- None: the device refused the member, e.g. several members or a damaged stream --,
    stats dict).  Test / benchmark entry: the ingest keeps the inflated bytes in HBM and never brings them back."""
    import numpy as np
    paths = [str(p) for p in paths]
    if capacity is None:
        capacity = 64
        for p in paths:                                           # ISIZE of the last member's trailer (mod 2^32)
            with open(p, "rb") as fh:
                fh.seek(-4, 2)
                capacity += int.from_bytes(fh.read(4), "little") + 64
    out = np.empty(int(capacity), dtype=np.uint8)
    lens = (C.c_uint64 * max(len(paths), 1))()
    stats = (C.c_double * 16)()
    arr = (C.c_char_p * max(len(paths), 1))(*[p.encode("utf-8") for p in paths])
    rustcall(lib.smgpu_gunzip_files, arr, len(paths), out.ctypes.data_as(C.POINTER(C.c_uint8)), int(capacity), lens, stats)
    res, at = [], 0
    for i in range(len(paths)):
        if lens[i] == 2**64 - 1:
            res.append(None)
        else:
            res.append(out[at:at + lens[i]].tobytes())
            at += lens[i]
    keys = ("survivors", "candidates", "runs", "scan_ms", "pass1_ms", "link_ms", "pass2_ms", "finish_ms", "device_total_ms", "io_ms")
    return res, dict(zip(keys, (float(v) for v in stats)))


def gunzip_counters():
    "(gzip files the ingest inflated on the device, files it handed to the host inflater after the device refused them)"
    out = (C.c_uint64 * 2)()
    lib.smgpu_gunzip_counters(out)
    return int(out[0]), int(out[1])
