#!/bin/bash
# tools/build_all_variant.sh NAME "-DFLAG=..."  -> tools/bin/lib_NAME.so (EVERY translation unit rebuilt with the extra flags)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/bin/obj_$NAME
OBJS=""
for src in humanvid_amd/csrc/*.hip humanvid_amd/csrc/hv_api.cpp; do
  b=$(basename $src); b=${b%.*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ihumanvid_amd/csrc "$@" -x hip -c $src -o tools/bin/obj_$NAME/$b.o 2>/dev/null &
  OBJS="$OBJS tools/bin/obj_$NAME/$b.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/bin/lib_$NAME.so
echo built tools/bin/lib_$NAME.so
