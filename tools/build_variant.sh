#!/bin/bash
# A variant library with ONE source compiled with extra flags, everything else from the current build:
#   bash tools/build_variant.sh csrc/conv_halo.hip nopf -DHALO_PREFETCH=0   ->  tools/variants/libltxhip_nopf.so   (select with LTXHIP_LIB)
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/candle-video_amd; SRC=$1; NAME=$2; shift 2
mkdir -p $R/tools/variants $P/build/var
make -C $P -j8 > /dev/null
OBJ=$P/build/var/$(basename $SRC).$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I$R/include "$@" -x hip -c $P/$SRC -o $OBJ
OBJS=$(find $P/build/csrc $P/build/host -name "*.o" | grep -v "/$(basename $SRC).o" | sort)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libltxhip_$NAME.so $OBJS $OBJ -lz -ldl
echo built tools/variants/libltxhip_$NAME.so
