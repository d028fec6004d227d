"""Exception classes keyed by the C-ABI's numeric error codes.

Same mapping rule as the reference (src/sourmash/exceptions.py:136-153): codes
strictly between 100 and 10000, and 1104, surface as ``ValueError``; every other
code gets a ``SourmashError`` subclass named after its SOURMASH_ERROR_CODE_*
constant (e.g. ``Internal``, ``SerdeError``), with ``XError`` families sharing a
common base.
"""
from ._lowlevel import lib

__all__ = ["SourmashError", "exceptions_by_code"]


class SourmashError(Exception):
    code = None

    def __init__(self, msg):
        super().__init__(msg)
        self.message = msg
        self.rust_info = None

    def __str__(self):
        return self.message if self.rust_info is None else f"{self.message}\n\n{self.rust_info}"


exceptions_by_code = {}


def _family_base(name):
    head, sep, tail = name.partition("Error")
    if sep and head and tail:
        base_name = head + "Error"
        base = globals().get(base_name)
        if base is None:
            base = type(base_name, (SourmashError,), {})
            globals()[base_name] = base
            __all__.append(base_name)
        return base
    return SourmashError


for _const, _code in sorted(lib.constants.items()):
    if not _const.startswith("SOURMASH_ERROR_CODE_"):
        continue
    if _code == 1104 or 100 <= _code <= 10000:
        exceptions_by_code[_code] = ValueError
        continue
    _name = _const[len("SOURMASH_ERROR_CODE_"):].title().replace("_", "")
    _cls = type(_name, (_family_base(_name),), {"code": _code})
    globals()[_name] = _cls
    __all__.append(_name)
    exceptions_by_code[_code] = _cls
