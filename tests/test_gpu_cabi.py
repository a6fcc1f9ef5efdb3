"""The C ABI without Python in the process: tools/hwcheck.cpp (built by __graft_entry__.build() into tools/bin/hwcheck)
links libhumanvid_hip.so through include/humanvid_hip.h alone, runs hv_groupnorm_affine against a double-precision host
reference (bench shape + edge shapes: concat seam, an empty pixel range, 80 channels per group) and hv_gemm under its kernel
selections (HV_TUNE_GEMM_GLDS 1 / 2 / 3 / 0) at the level-2 / level-3 projection shapes: the selections must agree bit for
bit, and sampled rows are checked against the host."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer_of_the_abi():
    exe = os.path.join(REPO, "tools", "bin", "hwcheck")
    if not os.path.exists(exe):
        from humanvid_amd import build as hvbuild

        hvbuild.build()
        exe = hvbuild.build_hwcheck()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    assert res.returncode == 0 and "hwcheck: all OK" in res.stdout, res.stdout + res.stderr
    assert res.stdout.count(" OK") >= 7 and "FAIL" not in res.stdout
