#!/bin/bash
# session 9: issue rates of the softmax instructions (alone, beside MFMAs, across waves of a SIMD) and the vendor library's
# rate on the step's GEMM shapes
mkdir -p gpurun_out
{
echo "== VALU / MFMA issue rates"
timeout 120 tools/bin/valu_rate
echo "== vendor library on the step's GEMM shapes"
timeout 300 python tools/lib_gemm_reference.py
echo "== own kernels on the same shapes"
timeout 300 python tools/microbench.py --only gemm 2>&1 | tail -30
} > gpurun_out/r04_s9.txt 2>&1
