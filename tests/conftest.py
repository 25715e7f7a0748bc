import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
MODES = [("zeros", False), ("zeros", True), ("border", False), ("border", True)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # MSDA_TEST_OPTS="value_path=2,q_round=11": run the suite under library option overrides (variant matrix)
    opts = os.environ.get("MSDA_TEST_OPTS", "")
    if opts:
        from msda_triton_amd import _lib
        for kv in opts.split(","):
            k, v = kv.split("=")
            _lib.set_option(k.strip(), int(v))


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU: the gpu-marked tests are skipped, not failed (the driver's own runs
    select with -m gpu / -m "not gpu" and are unaffected)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (marked gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def mode_key(pm, ac):
    return f"{pm}_{int(ac)}"


def golden_cases(dtype_tag=None):
    files = sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                   if not os.path.basename(f).startswith(("digest_", "module_")))
    if dtype_tag:
        files = [f for f in files if f.endswith(f"_{dtype_tag}.npz")]
    return files


def module_cases(dtype_tag=None):
    """fixtures of the reference's nn.Module (tests/golden/make_golden.py: write_module_cases)"""
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, "module_*.npz")))
    if dtype_tag:
        files = [f for f in files if f.endswith(f"_{dtype_tag}.npz")]
    return files


def load_module_case(path, device="cpu"):
    """-> (module built from the fixture's state dict, inputs dict, expected dict) as torch objects on `device`"""
    import torch
    from msda_triton_amd import MultiscaleDeformableAttention
    z = np.load(path)
    emb, hidden, L, H, P, B, Q, ref_dim, zeros, ac = (int(v) for v in z["meta"])
    dt = torch.from_numpy(z["img"]).dtype
    m = MultiscaleDeformableAttention(emb, hidden, L, H, P, "zeros" if zeros else "border", bool(ac)).to(dt)
    m.load_state_dict({k[len("param__"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param__")})
    m = m.to(device)
    inputs = {k: torch.from_numpy(z[k]).to(device) for k in ("img", "shapes", "queries", "reference_points", "grad_out")}
    expected = {k: torch.from_numpy(z[k]) for k in z.files
                if k in ("out", "grad_img", "grad_queries", "grad_reference_points") or k.startswith("grad__")}
    return m, inputs, expected


def digest_cases():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "digest_*.npz")))


def case_id(path):
    return os.path.basename(path)[:-4]


def kink_mask(loc, shapes, align_corners, tol=1e-4):
    """True where a sample's pixel-space x (resp. y) coordinate is within ``tol`` of an integer: there
    the location gradient is discontinuous and float round-off legitimately picks either side.
    Returns a bool array shaped like loc ([..., L, P, 2])."""
    loc = np.asarray(loc, dtype=np.float64)
    shapes = np.asarray(shapes)
    L = shapes.shape[0]
    size = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)  # (w, h) per level -> x, y
    size = size.reshape((1,) * (loc.ndim - 3) + (L, 1, 2))
    pix = loc * (size - 1) if align_corners else loc * size - 0.5
    return np.abs(pix - np.round(pix)) < tol


@pytest.fixture(scope="session")
def oracle():
    from oracle import msda_oracle
    msda_oracle.build()
    return msda_oracle
