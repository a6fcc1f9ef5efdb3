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
$H -Iinclude -Ihumanvid_amd/csrc -DHV_GEMM_TRACE=0 tools/gemm_trace.hip -o tools/bin/gemm_trace
$H -Iinclude -Ihumanvid_amd/csrc -DHV_GEMM_TRACE=0 tools/attn_trace.hip -o tools/bin/attn_trace
ls -la tools/bin
