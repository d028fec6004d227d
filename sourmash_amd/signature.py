"""SourmashSignature + JSON load/save over libsourmash_amd.so.

API of src/sourmash/signature.py:29-527 (class SourmashSignature,
FrozenSourmashSignature, load_signatures_from_json, load_one_signature_from_json,
save_signatures_to_json): a signature is metadata (name, filename, license) around
one or more sketches; ``.minhash`` hands back a frozen copy of the first
sketch; ``add_sequence`` fans a record out to every sketch on the GPU.
"""
import contextlib
import ctypes as C
import os

from ._lowlevel import ffi, lib
from .minhash import FrozenMinHash, MinHash, to_bytes
from .utils import RustObject, decode_str, rustcall

__all__ = ["SourmashSignature", "FrozenSourmashSignature", "load_signatures_from_json",
           "load_one_signature_from_json", "save_signatures_to_json"]

SIGNATURE_VERSION = 0.4


class SourmashSignature(RustObject):
    "Metadata + sketch(es)."

    __dealloc_func__ = lib.signature_free

    def __init__(self, minhash, name="", filename=""):
        self._objptr = lib.signature_new()
        if name:
            self.name = name
        if filename:
            self.filename = filename
        self.minhash = minhash

    # ---- sketch access --------------------------------------------------------------------
    @property
    def minhash(self):
        "A frozen COPY of the first sketch (ffi/signature.rs:167-182 clones)."
        return FrozenMinHash._from_objptr(self._methodcall(lib.signature_first_mh))

    @minhash.setter
    def minhash(self, value):
        self._methodcall(lib.signature_set_mh, value._get_objptr())

    def minhashes(self):
        "Copies of every sketch held (signature_get_mhs)."
        size = ffi.new_size()
        arr = self._methodcall(lib.signature_get_mhs, C.byref(size))
        try:
            return [FrozenMinHash._from_objptr(arr[i]) for i in range(size.value)]
        finally:
            lib.nodegraph_buffer_free(C.cast(arr, C.POINTER(C.c_uint8)), size.value * C.sizeof(C.c_void_p))

    def __len__(self):
        return self._methodcall(lib.signature_len)

    def md5sum(self):
        "md5 of the first sketch."
        mh = self.minhash
        return decode_str(mh._methodcall(lib.kmerminhash_md5sum))

    def __hash__(self):
        return hash(self.md5sum())

    def __eq__(self, other):
        return self._methodcall(lib.signature_eq, other._get_objptr())

    def __ne__(self, other):
        return not self == other

    # ---- metadata ----------------------------------------------------------------------------
    @property
    def name(self):
        return decode_str(self._methodcall(lib.signature_get_name))

    @name.setter
    def name(self, value):
        self._methodcall(lib.signature_set_name, to_bytes(value))

    _name = name

    @property
    def filename(self):
        return decode_str(self._methodcall(lib.signature_get_filename))

    @filename.setter
    def filename(self, value):
        self._methodcall(lib.signature_set_filename, to_bytes(value))

    @property
    def license(self):
        return decode_str(self._methodcall(lib.signature_get_license))

    def _display_name(self, max_length=0):
        name = self.name
        if name:
            if max_length and len(name) > max_length:
                name = name[:max_length - 3] + "..."
        elif self.filename:
            name = self.filename
            if max_length and len(name) > max_length:
                name = "..." + name[-max_length + 3:]
        else:
            name = self.md5sum()[:8]
        return name

    def __str__(self):
        return self._display_name()

    def __repr__(self):
        name, md5pref = self.name, self.md5sum()[:8]
        return f"SourmashSignature({md5pref})" if name == md5pref else f"SourmashSignature('{name}', {md5pref})"

    # ---- comparisons: delegate to the first sketch -------------------------------------------
    def similarity(self, other, ignore_abundance=False, downsample=False):
        return self.minhash.similarity(other.minhash, ignore_abundance=ignore_abundance, downsample=downsample)

    def jaccard(self, other):
        return self.minhash.similarity(other.minhash, ignore_abundance=True, downsample=False)

    def jaccard_ani(self, other, *, downsample=False, jaccard=None, prob_threshold=1e-3, err_threshold=1e-4):
        return self.minhash.jaccard_ani(other.minhash, downsample=downsample, jaccard=jaccard,
                                        prob_threshold=prob_threshold, err_threshold=err_threshold)

    def contained_by(self, other, downsample=False):
        return self.minhash.contained_by(other.minhash, downsample)

    def containment_ani(self, other, *, downsample=False, containment=None, confidence=0.95, estimate_ci=False):
        return self.minhash.containment_ani(other.minhash, downsample=downsample, containment=containment,
                                            confidence=confidence, estimate_ci=estimate_ci)

    def max_containment(self, other, downsample=False):
        return self.minhash.max_containment(other.minhash, downsample)

    def max_containment_ani(self, other, *, downsample=False, max_containment=None, confidence=0.95,
                            estimate_ci=False):
        return self.minhash.max_containment_ani(other.minhash, downsample=downsample,
                                                max_containment=max_containment, confidence=confidence,
                                                estimate_ci=estimate_ci)

    def avg_containment(self, other, downsample=False):
        return self.minhash.avg_containment(other.minhash, downsample=downsample)

    def avg_containment_ani(self, other, *, downsample=False):
        return self.minhash.avg_containment_ani(other.minhash, downsample=downsample)

    # ---- sketching -------------------------------------------------------------------------------
    def add_sequence(self, sequence, force=False):
        "Add one record to every sketch of the signature (GPU)."
        self._methodcall(lib.signature_add_sequence, to_bytes(sequence), force)

    def add_protein(self, sequence):
        self._methodcall(lib.signature_add_protein, to_bytes(sequence))

    @staticmethod
    def from_params(params):
        return SourmashSignature._from_objptr(rustcall(lib.signature_from_params, params._get_objptr()))

    # ---- copying / pickling -------------------------------------------------------------------
    def __getstate__(self):
        return (self.minhash, self.name, self.filename)

    def __setstate__(self, tup):
        mh, name, filename = tup
        self.__del__()
        self._shared = False
        self._objptr = lib.signature_new()
        if name:
            SourmashSignature.name.fset(self, name)
        if filename:
            SourmashSignature.filename.fset(self, filename)
        SourmashSignature.minhash.fset(self, mh)

    def __reduce__(self):
        return (SourmashSignature, (self.minhash, self.name, self.filename))

    def __copy__(self):
        return SourmashSignature(self.minhash, name=self.name, filename=self.filename)

    copy = __copy__

    def to_frozen(self):
        new_ss = self.copy()
        new_ss.__class__ = FrozenSourmashSignature
        return new_ss

    def to_mutable(self):
        return self.copy()

    def into_frozen(self):
        self.__class__ = FrozenSourmashSignature


def _frozen(*_a, **_k):
    raise ValueError("cannot modify frozen signature; use to_mutable() first")


class FrozenSourmashSignature(SourmashSignature):
    "Read-only signature (src/sourmash/signature.py:286-347)."
    minhash = property(SourmashSignature.minhash.fget, _frozen)
    name = property(SourmashSignature.name.fget, _frozen)
    _name = name
    filename = property(SourmashSignature.filename.fget, _frozen)
    add_sequence = add_protein = _frozen

    def __copy__(self):
        return self

    copy = __copy__

    def to_frozen(self):
        return self

    def to_mutable(self):
        mut = SourmashSignature.__new__(SourmashSignature)
        mut.__setstate__(self.__getstate__())
        return mut

    def into_frozen(self):
        pass

    @contextlib.contextmanager
    def update(self):
        "`with sig.update() as s:` -> mutable copy, frozen again on exit."
        new_copy = self.to_mutable()
        yield new_copy
        new_copy.into_frozen()


# ---- JSON -------------------------------------------------------------------------------------
def _as_buffer(data):
    "file-like / path / str / bytes -> bytes or None"
    if hasattr(data, "__fspath__") or (hasattr(data, "strpath") and not isinstance(data, (str, bytes))):
        data = os.fspath(data) if hasattr(data, "__fspath__") else data.strpath
    if hasattr(data, "read"):
        if hasattr(data, "mode") and "t" in data.mode and hasattr(data, "buffer"):
            data = data.buffer
        buf = data.read()
        data.close()
        data = buf
    if isinstance(data, str):
        if data.find("sourmash_signature") > 0:
            return data.encode("utf-8")
        if os.path.exists(data):
            with open(data, "rb") as fh:
                return fh.read()
        return None
    if isinstance(data, (bytes, bytearray, memoryview)):
        data = bytes(data)
        if data.find(b"sourmash_signature") > 0 or data.startswith(b"\x1f\x8b"):
            return data
        try:
            if os.path.exists(data):
                with open(data, "rb") as fh:
                    return fh.read()
        except (ValueError, TypeError):
            pass
    return None


def load_signatures_from_json(data, ksize=None, select_moltype=None, ignore_md5sum=False, do_raise=False):
    """Yield frozen signatures from JSON text / bytes (optionally gzip) / a path /
    a file object; one signature per sketch, filtered by ksize and moltype."""
    ksize = int(ksize) if ksize is not None else 0
    if not data:
        return
    buf = _as_buffer(data)
    if buf is None:
        if do_raise:
            raise ValueError("Error in parsing signature; quitting. Cannot open file or invalid signature")
        return
    mol = None if select_moltype is None else to_bytes(select_moltype)
    try:
        size = ffi.new_size()
        arr = rustcall(lib.signatures_load_buffer, buf, len(buf), ignore_md5sum, ksize, mol, C.byref(size))
        sigs = [SourmashSignature._from_objptr(arr[i]) for i in range(size.value)]
        lib.nodegraph_buffer_free(C.cast(arr, C.POINTER(C.c_uint8)), size.value * C.sizeof(C.c_void_p))
        for sig in sigs:
            yield sig.to_frozen()
    except Exception:
        if do_raise:
            raise


def load_one_signature_from_json(data, ksize=None, select_moltype=None, ignore_md5sum=False):
    it = load_signatures_from_json(data, ksize=ksize, select_moltype=select_moltype, ignore_md5sum=ignore_md5sum)
    try:
        first = next(it)
    except StopIteration:
        raise ValueError("no signatures to load")
    try:
        next(it)
    except StopIteration:
        return first
    raise ValueError("expected to load exactly one signature")


def save_signatures_to_json(siglist, fp=None, compression=0):
    "Serialise signatures to JSON bytes (gzip when compression > 0; src/sourmash/signature.py:493-527), or write to `fp`."
    sigs = list(siglist)
    ptrs = (C.c_void_p * max(len(sigs), 1))(*[s._get_objptr() for s in sigs])
    size = ffi.new_size()
    raw = rustcall(lib.signatures_save_buffer, ptrs, len(sigs), compression, C.byref(size))
    try:
        result = C.string_at(raw, size.value)
    finally:
        lib.nodegraph_buffer_free(raw, size.value)
    if fp is None:
        return result
    try:
        fp.write(result)
    except TypeError:                                    # a text-mode handle
        fp.write(result.decode("utf-8"))
    return None
