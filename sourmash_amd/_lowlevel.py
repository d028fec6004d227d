"""ctypes binding of libsourmash_amd.so -- the stand-in for the reference's
``sourmash._lowlevel`` cffi module (``from ._lowlevel import ffi, lib``;
pyproject.toml:138-155, header include/sourmash.h).

cffi is not available in the target image, so the prototypes are derived at
import time by parsing ``include/sourmash_amd.h`` (the same header a cffi build
would consume) and attached to the shared library with ctypes.  ``lib`` exposes
every declared function plus the enum constants; ``ffi`` offers the handful of
cffi helpers the reference's Python layer uses (new / unpack / string / NULL).

The library is located in-tree (``sourmash_amd/libsourmash_amd.so``, built by
``__graft_entry__.build()`` / ``make -C sourmash_amd/csrc``).  Import torch first
if you use it in the same process so both share one HIP runtime.
"""
import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "sourmash_amd.h")
# SMG_LIBRARY: another build of the same library (the sanitizer build of tests/test_sanitizers_cpu.py)
LIBPATH = os.environ.get("SMG_LIBRARY") or os.path.join(_PKG, "libsourmash_amd.so")


class SourmashStr(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_char)), ("len", C.c_size_t), ("owned", C.c_bool)]


_SCALARS = {
    "void": None, "bool": C.c_bool, "char": C.c_char, "double": C.c_double,
    "uint8_t": C.c_uint8, "uint32_t": C.c_uint32, "int32_t": C.c_int32, "uint64_t": C.c_uint64,
    "uintptr_t": C.c_size_t, "HashFunctions": C.c_uint32, "SourmashErrorCode": C.c_uint32,
    "SourmashStr": SourmashStr,
}
_OPAQUE = {"SourmashKmerMinHash", "SourmashSignature", "SourmashComputeParameters", "SmgpuSketchSet", "SmgpuCounter", "SmgpuBitIndex",
           "SmgpuGather", "SmgpuCollection", "SmgpuGatherXchg"}


def _ctype(decl, is_arg=False):
    """C type text (without the parameter name) -> ctypes type.  Data-pointer ARGUMENTS become
    c_void_p so that numpy buffers, ctypes arrays, byref() and raw device addresses all pass."""
    t = decl.replace("const", " ").strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()[0]
    if is_arg and stars and not (base == "char" and stars == 1):
        return C.c_void_p
    if base in _OPAQUE:
        return C.c_void_p if stars == 1 else C.POINTER(C.c_void_p)
    if stars == 0:
        return _SCALARS[base]
    if base == "char" and stars == 1:
        return C.c_char_p
    if base == "void":
        return C.c_void_p
    if base == "SourmashStr":
        return C.POINTER(SourmashStr)
    ct = _SCALARS[base]
    for _ in range(stars):
        ct = C.POINTER(ct)
    return ct


def parse_header(path=HEADER):
    """-> (functions {name: (restype_text, [argtype_text...])}, constants {name: int})"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    consts = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        for name, val in re.findall(r"(\w+)\s*=\s*(\d+)", body):
            consts[name] = int(val)
    text = re.sub(r"enum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", " ", text)
    text = text.replace('extern "C" {', " ")
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if not ret:
            continue
        argl = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)      # strip the parameter name
                argl.append(mm.group(1).strip() if mm and mm.group(1).strip() else a)
        funcs[name] = (ret, argl)
    return funcs, consts


class _Lib:
    """Lazy handle: parsing the header is cheap, loading the .so happens once."""

    def __init__(self):
        self._cdll = None
        self.functions, self.constants = parse_header()
        for k, v in self.constants.items():
            setattr(self, k, v)

    def _load(self):
        if self._cdll is None:
            if not os.path.exists(LIBPATH) and os.path.exists("/opt/rocm/bin/hipcc"):
                # first use in a fresh checkout: build the HIP library in-tree (same recipe as __graft_entry__.build)
                import subprocess
                import sys
                print("sourmash_amd: building libsourmash_amd.so (make -C sourmash_amd/csrc) ...", file=sys.stderr)
                subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-j8", "-s"])
            if not os.path.exists(LIBPATH):
                raise ImportError(
                    f"{LIBPATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C sourmash_amd/csrc` (there is no pure-Python / CPU fallback)")
            # Share ONE HIP runtime with PyTorch: torch bundles its own libamdhip64 (same SONAME as
            # /opt/rocm's).  Loaded first, the dynamic linker binds our NEEDED libamdhip64.so.7 to it;
            # loaded second, the process would end up with two HIP runtimes.
            try:
                import torch  # noqa: F401
            except Exception:  # torch absent: stand-alone use with the system ROCm runtime
                pass
            self._cdll = C.CDLL(LIBPATH, mode=C.RTLD_GLOBAL)
            for name, (ret, args) in self.functions.items():
                fn = getattr(self._cdll, name)       # AttributeError here == missing export
                fn.restype = _ctype(ret)
                fn.argtypes = [_ctype(a, is_arg=True) for a in args]
            self._cdll.sourmash_init()
        return self._cdll

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = getattr(self._load(), name)
        setattr(self, name, fn)
        return fn


lib = _Lib()


class _FFI:
    """The slice of cffi's `ffi` object used by the reference's Python layer."""
    NULL = None

    @staticmethod
    def new_size():
        return C.c_size_t(0)

    @staticmethod
    def new_u64():
        return C.c_uint64(0)

    @staticmethod
    def unpack_u64(ptr, n):
        return [ptr[i] for i in range(n)] if n else []

    @staticmethod
    def u64_array(values):
        values = list(values)
        return (C.c_uint64 * max(len(values), 1))(*values), len(values)

    @staticmethod
    def string(s):
        """SourmashStr -> python str"""
        if not s.data or not s.len:
            return ""
        return C.string_at(s.data, s.len).decode("utf-8")


ffi = _FFI()


def decode_str(s):
    """SourmashStr by value -> str, freeing it when owned (src/sourmash/utils.py:41-49)."""
    try:
        return ffi.string(s)
    finally:
        if s.owned:
            lib.sourmash_str_free(C.byref(s))
