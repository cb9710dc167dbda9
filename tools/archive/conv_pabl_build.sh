#!/bin/bash
# Variant libraries of the pipelined conv_halo form with timing ablations (HALO_PABL, csrc/conv_halo.hip): tools/variants/libltxhip_halo_pN.so
# run in the build container after `make -C candle-video_amd`; measured by tools/conv_variants.py on the GPU box.
set -e
cd "$(dirname "$0")/../candle-video_amd"
mkdir -p build/var ../tools/variants
for P in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DHALO_PABL=$P -x hip -c csrc/conv_halo.hip -o build/var/conv_halo_p$P.o
  objs=$(ls build/csrc/*.o build/host/*.o | grep -v "conv_halo.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/variants/libltxhip_halo_p$P.so $objs build/var/conv_halo_p$P.o -lz -ldl
  echo built halo_p$P
done
