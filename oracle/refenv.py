"""Enter the reference's import environment  --  TEST INFRASTRUCTURE ONLY (build container only).

`enter()` makes `import src.…` resolve to /root/reference (the repo's own drop-in surface is also
called `src`, so the caller must be its own process), installs oracle/refshim (stand-ins for the
un-vendored diffusers==0.24.0) and returns (shim module, oracle_torch module).  Used by
oracle/gen_fullsize_golden.py and oracle/gen_refnet_golden.py; oracle/gen_golden.py keeps its own
copy of these few lines.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HV_REFERENCE_ROOT", "/root/reference")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def present() -> bool:
    return os.path.isdir(REF)


def enter():
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    sys.path.insert(0, REF)
    shim = _load("hv_refshim", os.path.join(REPO, "oracle", "refshim", "shim.py"))
    shim.install()
    O = _load("hv_oracle_torch", os.path.join(REPO, "oracle", "oracle_torch.py"))
    return shim, O


def stub_dataset_imports():
    """src/dataset/dance_image_h_v_camera.py imports torchvision / decord / transformers for its Dataset
    classes; Camera and ray_condition need none of them."""
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "decord",
                 "src.dataset.visualization_utils", "transformers"):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(VideoReader=None, CameraPoseVisualizer=None, visualize_camera_pose=None, to_image=None,
                          pca_visualize=None, CLIPImageProcessor=None)
        sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
