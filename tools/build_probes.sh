#!/bin/bash
# Build the standalone probes into tools/bin/ (git-ignored; travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result"
$H tools/fillbw.hip -o tools/bin/fillbw
$H tools/storebw.hip -o tools/bin/storebw
$H tools/valubw.hip -o tools/bin/valubw
$H tools/valu_rate.hip -o tools/bin/valu_rate
$H tools/dmabw.hip -o tools/bin/dmabw
if [ "$1" == "--trace" ]; then
  # the phase-trace tools need the s_memtime marks, which are NOT in the product sources: build them from a patched scratch copy
  T=$(mktemp -d); mkdir -p $T/humanvid_amd; cp -r humanvid_amd/csrc $T/humanvid_amd/; (cd $T && patch -s -p1 < "$OLDPWD/tools/trace_marks.patch")
  $H -Iinclude -I$T/humanvid_amd/csrc -DHV_GEMM_TRACE=0 tools/gemm_trace.hip -o tools/bin/gemm_trace
  $H -Iinclude -I$T/humanvid_amd/csrc -DHV_GEMM_TRACE=0 tools/attn_trace.hip -o tools/bin/attn_trace
  rm -rf $T
fi
ls -la tools/bin
