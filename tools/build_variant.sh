#!/bin/bash
# tools/build_variant.sh NAME UNIT "-DFLAG=..."  -> tools/bin/lib_NAME.so (one TU rebuilt with extra flags, rest reused)
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift 2
mkdir -p tools/bin/obj_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ihumanvid_amd/csrc "$@" -x hip -c humanvid_amd/csrc/$UNIT.hip -o tools/bin/obj_$NAME/$UNIT.o 2>/dev/null
OBJS=""
for o in humanvid_amd/lib/obj/*.o; do b=$(basename $o); if [ "$b" == "$UNIT.o" ]; then OBJS="$OBJS tools/bin/obj_$NAME/$UNIT.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/bin/lib_$NAME.so
echo built tools/bin/lib_$NAME.so
