#!/bin/bash
# Build a library variant for tools/ab.sh: tools/build_variant.sh <name> "<extra hipcc flags>"
# (the default build is left in place afterwards)
set -e
name=$1; flags=$2
mkdir -p chsimpy_amd/lib/variants
cp chsimpy_amd/lib/libchs_hip.so /tmp/libchs_default.so 2>/dev/null || true
CHS_EXTRA_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build_hip(force=True)"
cp chsimpy_amd/lib/libchs_hip.so chsimpy_amd/lib/variants/$name.so
if [ -f /tmp/libchs_default.so ]; then cp /tmp/libchs_default.so chsimpy_amd/lib/libchs_hip.so; touch chsimpy_amd/lib/libchs_hip.so; fi
echo "built variants/$name.so"
