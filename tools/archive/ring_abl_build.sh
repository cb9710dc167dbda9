#!/bin/bash
# Variant libraries of gemm_ring.hip with timing ablations (RING_ABL: 1 no MFMAs, 2 no fragment reads, 4 every piece re-reads one
# 1-KiB line set): tools/variants/libltxhip_ring_aN.so; run after `make -C candle-video_amd`, measured by tools/ring_abl.py.
set -e
cd "$(dirname "$0")/../candle-video_amd"
mkdir -p build/var ../tools/variants
for P in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DRING_ABL=$P -x hip -c csrc/gemm_ring.hip -o build/var/gemm_ring_a$P.o
  objs=$(ls build/csrc/*.o build/host/*.o | grep -v "gemm_ring.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/variants/libltxhip_ring_a$P.so $objs build/var/gemm_ring_a$P.o -lz -ldl
  echo built ring_a$P
done
