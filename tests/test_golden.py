"""The oracle against the committed golden vectors (tests/golden/, made by scripts/make_golden.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_golden as mg  # noqa: E402
import parity_utils as pu  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def test_oracle_ibl_matches_golden():
    gold = np.load(os.path.join(GOLD, "ibl_small.npz"))
    s = pu.SmallScene.build(gpu=False, **mg.FRAME_CFG)
    m = s.oracle.maps()
    np.testing.assert_array_equal(m["sky"], gold["sky"])
    np.testing.assert_allclose(m["irr"], gold["irr"], rtol=1e-5)
    for i, p in enumerate(m["pre"]):
        np.testing.assert_allclose(p, gold[f"pre{i}"], rtol=1e-5)
    a, b = m["lut"].view(np.float16).astype(np.float32), gold["lut"].view(np.float16).astype(np.float32)
    assert (np.isnan(a) == np.isnan(b)).all()
    np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], atol=1e-3)


def test_oracle_frames_match_golden():
    gold = np.load(os.path.join(GOLD, "frame_160x90.npz"))
    s = pu.SmallScene.build(gpu=False, **mg.FRAME_CFG)
    for name, moves in mg.VIEWS.items():
        s.renderer.camera = s.pkg.Camera()
        mg.apply_view(s.renderer.camera, moves)
        rgba, vis = s.render_oracle()
        assert (vis == gold[f"{name}_vis"]).mean() >= 0.9999, name
        d = np.abs(rgba.astype(np.int32) - gold[f"{name}_rgba"].astype(np.int32)).max(-1)
        assert (d <= 1).mean() >= 0.999, name
        assert (vis > 0).mean() > 0.02, name
