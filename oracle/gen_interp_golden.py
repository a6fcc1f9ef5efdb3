"""Golden vectors for the latent interpolation post-step  --  TEST INFRASTRUCTURE (build container only: needs /root/reference).

Runs the REFERENCE's own code: `interpolate_latents` is cut out of /root/reference/src/pipelines/pipeline_pose2vid_long.py
with `ast` (the module itself imports diffusers, which is not installed) and executed against the reference's own
/root/reference/src/pipelines/utils.py (pure torch, imported as is).  Nothing of the reference is written into the repo
but the produced numbers: tests/golden/latent_interp.npz.

    python oracle/gen_interp_golden.py            # regenerate
    python oracle/gen_interp_golden.py --check    # compare the oracle restatement with the reference, no write
"""
import argparse
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HV_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden", "latent_interp.npz")


def reference_functions():
    spec = importlib.util.spec_from_file_location("hv_ref_pipe_utils", os.path.join(REF, "src", "pipelines", "utils.py"))
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)
    path = os.path.join(REF, "src", "pipelines", "pipeline_pose2vid_long.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "interpolate_latents")
    ns = {"torch": torch, "get_tensor_interpolation_method": utils.get_tensor_interpolation_method}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return utils, ns["interpolate_latents"]


def cases():
    g = torch.Generator().manual_seed(7)
    a = torch.randn(1, 4, 5, 3, 2, generator=g)
    b = torch.randn(2, 4, 3, 2, 2, generator=g)
    c = torch.randn(1, 4, 4, 3, 2, generator=g)
    c[:, :, 2] = 1.0002 * c[:, :, 1] + 1e-4 * torch.randn(1, 4, 3, 2, generator=g)  # a nearly parallel pair: linear branch of slerp
    return {"a": a, "b": b, "c": c}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    utils, ref_interp = reference_functions()
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_torch as O

    out = {}
    worst = 0.0
    for name, lat in cases().items():
        out["in_" + name] = lat.numpy()
        for is_slerp in (False, True):
            utils.set_tensor_interpolation_method(is_slerp)
            for k in (1, 2, 3):
                ref = ref_interp(None, lat.clone(), k, "cpu")
                mine = O.interpolate_latents(lat.clone(), k, is_slerp)
                worst = max(worst, float((ref - mine).abs().max()))
                out[f"out_{name}_{'slerp' if is_slerp else 'linear'}_{k}"] = ref.numpy()
    print("oracle restatement vs the reference's own interpolate_latents: max |diff|", worst)
    assert worst <= 1e-6, worst
    if not args.check:
        np.savez_compressed(OUT, **out)
        print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
