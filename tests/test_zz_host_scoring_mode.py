"""The host mirror's scoring-mode switch (cfg key hip_lenet_mode, HipClassifier::setScoringMode): with the f32 chain selected
the CLI's scores are the oracle's, bit for bit; the default (split operands) stays within 1e-4 of them."""
import subprocess

import numpy as np
import pytest

from gpd_amd import synth
from test_host_cli import CLI, _subsample_indices, _write_case


@pytest.mark.gpu
def test_detect_grasps_cli_in_chain_mode_prints_the_oracles_scores(tmp_path, oracle_mod, lenet15_real):
    cl = synth.make_cloud(99, 12000)
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 150, 20, extra="hip_lenet_mode = 1\n")
    out = subprocess.run([CLI, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "hip_lenet_mode: 1" in out.stdout
    got = [l.split() for l in out.stdout.splitlines() if l.startswith("GRASP ")]
    assert len(got) == 20
    si = _subsample_indices(len(cl["xyz"]), 150)
    p = oracle_mod.default_params(15)
    hands, n, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, lenet15_real)
    v = hands[hands["valid"].astype(bool)]
    want = v[np.argsort(-v["score"], kind="stable")[:20]]
    gs = np.array([float(g[1]) for g in got]).astype(np.float32)  # printed with 9 significant digits: float32 round-trips
    assert np.array_equal(gs, want["score"].astype(np.float32)), np.abs(gs - want["score"]).max()
    assert [int(g[6]) for g in got] == want["finger_placement_index"].tolist()


def test_scoring_mode_key_is_documented():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "hip_lenet_mode" in open(os.path.join(root, "INTEGRATION.md")).read()
    assert "hip_lenet_mode" in open(os.path.join(root, "gpd_amd", "host", "src", "gpd_host.cpp")).read()


@pytest.mark.gpu
def test_graft_entry_smoke():
    """the driver's smoke() as a test of the GPU suite"""
    import __graft_entry__ as g
    g.smoke()
