#!/bin/bash
# Round 5, session 13: do two co-resident workgroups run k-loop and epilogue in LOCKSTEP?  Scratch builds (tools/bin/src_ph):
# the second workgroup to arrive on a CU (per-CU atomic counter, CU identity from HW_ID / XCC_ID) starts
# HV_PHASE_OFFSET x (K / 64) x 1024 cycles late.  ph0 = no offset (same sources), ph1..3 = offsets; applied to the 128x128
# kernel (two workgroups per CU) and to the archived hv_gemm_g2_kernel (tuning key 8 = 1).
# Build (here): S=tools/bin/src_ph = a copy of humanvid_amd/csrc/ + include/humanvid_hip.h with profiles/r05_gemm_g2.patch (csrc / include
#   hunks) and profiles/r05_phase_offset.patch applied; k_gemm.hip compiled with -DHV_PHASE_OFFSET=1|2|3 (ph0: without), hv_api.cpp
#   once, linked with the other objects of humanvid_amd/lib/obj/ into tools/bin/lib_gemm_ph{0..3}.so.
mkdir -p gpurun_out
OUT=gpurun_out/r05_s13.txt
{
for i in 1 2; do
for v in 0 1 2 3; do
echo "== product selection, lib_gemm_ph$v"
HV_LIB=tools/bin/lib_gemm_ph$v.so timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "gemm " | head -12
done
done
for v in 0 1 2 3; do
echo "== 128x128 kernel everywhere (key 3 = 3), lib_gemm_ph$v"
HV_TUNE="3=3" HV_LIB=tools/bin/lib_gemm_ph$v.so timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "gemm " | head -12
done
for v in 0 1 2 3; do
echo "== g2 kernel for the 256x256 problems (key 8 = 1), lib_gemm_ph$v"
HV_TUNE="8=1" HV_LIB=tools/bin/lib_gemm_ph$v.so timeout 300 python tools/microbench.py --only gemm 2>&1 | grep "gemm " | head -12
done
} > $OUT 2>&1
cat $OUT
