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
# ... and no 1024-thread kernel that is built to share its CU with a second workgroup (<= 64 vector registers: eight waves a SIMD)
# takes more than 80 scalar registers: k_mtf at 86 ("occupancy 8" to the compiler) computed wrong ranks on the device, differently
# from run to run (k_mtf.hip)
READELF=/opt/rocm/lib/llvm/bin/llvm-readelf
for co in "$D"/lib.so.*gfx950; do
  $READELF --notes "$co" 2>/dev/null | grep -E '\.name:|\.sgpr_count:|\.vgpr_count:|\.max_flat_workgroup_size:' | paste - - - - | while read -r line; do
    wg=$(echo "$line" | sed -n 's/.*max_flat_workgroup_size: *\([0-9]*\).*/\1/p'); sg=$(echo "$line" | sed -n 's/.*sgpr_count: *\([0-9]*\).*/\1/p')
    vg=$(echo "$line" | sed -n 's/.*vgpr_count: *\([0-9]*\).*/\1/p'); nm=$(echo "$line" | sed -n 's/.*\.name: *\([^ \t]*\).*/\1/p')
    if [ "${wg:-0}" = 1024 ] && [ "${vg:-999}" -le 64 ] && [ "${sg:-0}" -gt 80 ]; then echo "$nm: $sg scalar registers in a 1024-thread kernel of $vg vector registers"; touch "$D/bad_sgpr"; fi
  done
done
[ -e "$D/bad_sgpr" ] && bad=1
[ $n -gt 0 ] || { echo "no gfx950 code objects found in $LIB"; exit 2; }
[ $bad = 0 ] && echo "ok: $n gfx950 code objects, no calls"
exit $bad
