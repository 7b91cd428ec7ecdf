#!/bin/bash
# Build a library variant for tools/ab.sh: tools/build_variant.sh <name> "<extra hipcc flags>"
# The variant goes to chsimpy_amd/lib/variants/<name>.so (objects cached per variant); the product library
# chsimpy_amd/lib/libchs_hip.so is never touched.  A variant is selected through CHS_LIB_PATH.
set -e
name=$1; flags=$2
python - "$name" "$flags" <<'PY'
import sys
sys.path.insert(0, '.')
from chsimpy_amd import _build
out = _build.build_hip(out=f"{_build.LIBDIR}/variants/{sys.argv[1]}.so", extra=sys.argv[2])
print("built", out, _build.embedded_provenance(out))
PY
