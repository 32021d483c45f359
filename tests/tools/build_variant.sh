#!/bin/bash
# Ad-hoc tuning helper: build liblbzamd variants with extra -D flags into lbzip2_amd/csrc/variants/<name>.so
# usage: tests/tools/build_variant.sh name "-DMSD_BITS=32u ..."
set -e
cd /root/repo/lbzip2_amd/csrc
mkdir -p variants/$1
for f in k_collect k_bwt k_mtf k_encode k_finish k_decode lbz_api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-function -Wno-inline-asm $2 -c $f.hip -o variants/$1/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$1.so variants/$1/*.o
rm -rf variants/$1
echo built variants/$1.so
