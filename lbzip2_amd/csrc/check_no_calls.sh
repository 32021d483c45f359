#!/bin/bash
# The library's kernels contain no device function calls (k_bwt.hip, note at lds_radix_sort): fail if any gfx950 code object in
# the given shared library has an s_swappc / s_call.   usage: check_no_calls.sh liblbzamd.so
set -e
LIB=$(readlink -f "${1:-$(dirname "$0")/liblbzamd.so}")
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
D=$(mktemp -d); trap 'rm -rf "$D"' EXIT
cp "$LIB" "$D/lib.so"
(cd "$D" && $OBJDUMP --offloading lib.so > /dev/null)
n=0; bad=0
for co in "$D"/lib.so.*gfx950; do
  n=$((n + 1))
  c=$($OBJDUMP -d "$co" | grep -c -E "s_swappc|s_call_b64" || true)
  if [ "$c" != 0 ]; then echo "$(basename "$co"): $c call instruction(s)"; bad=1; fi
done
[ $n -gt 0 ] || { echo "no gfx950 code objects found in $LIB"; exit 2; }
[ $bad = 0 ] && echo "ok: $n gfx950 code objects, no calls"
exit $bad
