#!/bin/bash
# Stage the reference's own hot-path test modules next to the import shims (tests/refcompat/shim) in _refrun/
# (git-ignored, removed afterwards) and run them UNMODIFIED on a GPU box with `sourmash` = sourmash_amd.
# Runs in the build container (needs /root/reference); the summary lands in gpurun_out/reference_tests.txt.
#   tools/run_reference_tests.sh [test files...]      default: the modules of the hot path
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=/root/reference/tests
MODS=${@:-test_minhash.py test__minhash_hypothesis.py test_jaccard.py test_signature.py test_sketchcomparison.py test_search.py test_distance_utils.py test_compare.py test_index_protocol.py test_index.py test_api.py test_prefetch.py test_bugs.py test_deprecated.py test_sourmash_sketch.py}
rm -rf "$ROOT/_refrun"; mkdir -p "$ROOT/_refrun"
cp -r "$ROOT"/tests/refcompat/shim/* "$ROOT/_refrun/"
cp -r "$REF/test-data" "$ROOT/_refrun/test-data"
for m in $MODS; do cp "$REF/$m" "$ROOT/_refrun/"; done
chmod -R u+w "$ROOT/_refrun"     # (the reference tree is read-only: a test that rewrites a staged fixture would fail on its permissions)
trap 'rm -rf "$ROOT/_refrun"' EXIT
if [ -n "$SMG_REFRUN_LOCAL" ]; then      # collection / host-only check in the build container (no GPU: device calls fail)
  cd "$ROOT/_refrun" && PYTHONPATH=$ROOT python -m pytest -q -p no:cacheprovider --tb=short $MODS
else
  /usr/local/graft/bin/gpurun --timeout ${SMG_REFRUN_TIMEOUT:-900} -- "cd _refrun && PYTHONPATH=\$GRAFT_REPO_ROOT timeout 800 python -m pytest -q -p no:cacheprovider --tb=short -rfEs --junitxml=\$GRAFT_REPO_ROOT/gpurun_out/reference_tests.xml $MODS > \$GRAFT_REPO_ROOT/gpurun_out/reference_tests.txt 2>&1; tail -5 \$GRAFT_REPO_ROOT/gpurun_out/reference_tests.txt"
fi

# per-module table for profiles/ (the test sources themselves are never kept)
python - "$ROOT/gpurun_out/reference_tests.xml" <<'PY'
import collections, re, sys
import xml.etree.ElementTree as ET
if len(sys.argv) > 1:
    try:
        root = ET.parse(sys.argv[1]).getroot()
    except OSError:
        sys.exit(0)
    tab = collections.defaultdict(collections.Counter)
    why = collections.Counter()
    for tc in root.iter("testcase"):
        mod = tc.get("classname", "?")
        kind = "passed"
        for child in tc:
            if child.tag in ("failure", "error"):
                kind = "failed"
            elif child.tag == "skipped":
                kind = "skipped"
                why[(child.get("message") or "").split("\n")[0].replace("Skipped: ", "")[:120]] += 1
        tab[mod][kind] += 1
    print("%-32s %7s %7s %7s" % ("reference test module", "passed", "failed", "skipped"))
    tot = collections.Counter()
    for mod in sorted(tab):
        c = tab[mod]; tot.update(c)
        print("%-32s %7d %7d %7d" % (mod + ".py", c["passed"], c["failed"], c["skipped"]))
    print("%-32s %7d %7d %7d" % ("total", tot["passed"], tot["failed"], tot["skipped"]))
    print("\nskip reasons:")
    for k, v in why.most_common():
        print("%5d  %s" % (v, k))
PY
