"""`import sourmash` -> sourmash_amd (test harness only; see tests/refcompat/README.md)."""
import importlib
import sys

import sourmash_amd as _impl

for _name in ("minhash", "signature", "sketchcomparison", "search", "index", "compare", "distance_utils",
              "exceptions", "utils", "_lowlevel", "sketch"):
    _mod = importlib.import_module("sourmash_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod

sys.modules[__name__ + ".command_sketch"] = command_sketch = sketch   # noqa: F821  (one module here)

from sourmash_amd import *                                  # noqa: F401,F403,E402
from sourmash_amd import (MinHash, FrozenMinHash, SourmashSignature, load_signatures_from_json,  # noqa: E402
                          load_one_signature_from_json, save_signatures_to_json, DEFAULT_SEED, MAX_HASH)
from sourmash_amd.minhash import get_minhash_default_seed, get_minhash_max_hash  # noqa: F401,E402

VERSION = _impl.VERSION


import os as _os                                           # noqa: E402
import warnings as _warnings                                # noqa: E402


def _deprecated(what):
    _warnings.warn(what + " is deprecated", DeprecationWarning, stacklevel=3)


def load_signatures(*args, **kwargs):
    _deprecated("load_signatures")
    return load_signatures_from_json(*args, **kwargs)


def load_one_signature(*args, **kwargs):
    _deprecated("load_one_signature")
    return load_one_signature_from_json(*args, **kwargs)


def save_signatures(*args, **kwargs):
    _deprecated("save_signatures")
    return save_signatures_to_json(*args, **kwargs)



# ---- everything outside the hot path: importable, skips the test when touched -------------------
import types as _types                                      # noqa: E402


class _Skips(type):
    def __call__(cls, *args, **kwargs):
        import pytest
        pytest.skip(cls.__name__ + ": outside the hot path (SURVEY.md section 8)")

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Skips(cls.__name__ + "." + name, (), {})


class _OutOfScope(_types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Skips(self.__name__ + "." + name, (), {})


for _name in ("sbt", "sbtmh", "lca", "lca.lca_db", "lca.lca_utils", "index.sqlite_index", "index.revindex",
              "sourmash_args", "tax", "nodegraph", "hll", "cli", "cli.utils", "commands", "sig", "plugins", "sbt_storage", "logging",
              "manifest", "picklist", "save_load"):
    _mod = _OutOfScope(__name__ + "." + _name)
    sys.modules[_mod.__name__] = _mod
    if "." not in _name:
        globals()[_name] = _mod
sys.modules[__name__ + ".lca"].lca_db = sys.modules[__name__ + ".lca.lca_db"]
for _name in ("ZipFileLinearIndex", "LazyLinearIndex", "MultiIndex", "StandaloneManifestIndex"):
    if not hasattr(index, _name):                           # noqa: F821  (bound by the loop at the top)
        setattr(index, _name, _Skips("sourmash.index." + _name, (), {}))   # noqa: F821
create_sbt_index = _Skips("sourmash.create_sbt_index", (), {})
load_sbt_index = _Skips("sourmash.load_sbt_index", (), {})
search_sbt_index = _Skips("sourmash.search_sbt_index", (), {})


# collection loaders (zip / directory / manifest / path list -> Index objects) are the reference's control plane
load_file_as_index = _Skips("sourmash.load_file_as_index", (), {})


def load_file_as_signatures(filename, *, select_moltype=None, ksize=None, picklist=None, **kwargs):
    """Test plumbing: the reference's fixtures read single JSON signature files through its loader chain
    (sourmash_args.py:765-830), which as a whole is control plane.  A .sig / .sig.gz file goes through the JSON loader of
    the signature module; every other container (zip, directory, SBT, ...) skips the test."""
    name = str(filename)
    if picklist is not None or _os.path.isdir(name) or not name.endswith((".sig", ".sig.gz", ".json")):
        import pytest
        pytest.skip("sourmash.load_file_as_signatures on a collection: outside the hot path (SURVEY.md section 8)")
    return list(load_signatures_from_json(name, ksize=ksize, select_moltype=select_moltype))
