#!/bin/bash
# Stage the reference's own hot-path test modules next to the import shims (tests/refcompat/shim) in _refrun/
# (git-ignored, removed afterwards) and run them UNMODIFIED on a GPU box with `sourmash` = sourmash_amd.
# Runs in the build container (needs /root/reference); the summary lands in gpurun_out/reference_tests.txt.
#   tools/run_reference_tests.sh [test files...]      default: the modules of the hot path
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=/root/reference/tests
MODS=${@:-test_minhash.py test__minhash_hypothesis.py test_jaccard.py test_signature.py test_sketchcomparison.py test_search.py test_distance_utils.py test_compare.py}
rm -rf "$ROOT/_refrun"; mkdir -p "$ROOT/_refrun"
cp -r "$ROOT"/tests/refcompat/shim/* "$ROOT/_refrun/"
cp -r "$REF/test-data" "$ROOT/_refrun/test-data"
for m in $MODS; do cp "$REF/$m" "$ROOT/_refrun/"; done
trap 'rm -rf "$ROOT/_refrun"' EXIT
if [ "$1" = "--local" ] || [ -n "$SMG_REFRUN_LOCAL" ]; then
  cd "$ROOT/_refrun" && PYTHONPATH=$ROOT python -m pytest -q -p no:cacheprovider --tb=short $MODS
else
  /usr/local/graft/bin/gpurun --timeout ${SMG_REFRUN_TIMEOUT:-900} -- "cd _refrun && PYTHONPATH=\$GRAFT_REPO_ROOT timeout 800 python -m pytest -q -p no:cacheprovider --tb=short -rfEs $MODS > \$GRAFT_REPO_ROOT/gpurun_out/reference_tests.txt 2>&1; tail -5 \$GRAFT_REPO_ROOT/gpurun_out/reference_tests.txt"
fi
