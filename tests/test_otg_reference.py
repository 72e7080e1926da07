"""CPU tests of the trajectory-generation row: the committed fixtures tests/golden/otg_*.npz ARE outputs of the
reference (vendored Ruckig 0.10.1 compiled in place, oracle/_ref) -- when that build is present the fixtures are regenerated and
must match bit for bit; the invariants below need no second implementation."""
import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import make_otg_golden as G  # noqa: E402
import otg_ref  # noqa: E402

ALL = sorted(glob.glob(os.path.join(HERE, "golden", "otg_*.npz")))
FIX = [p for p in ALL if "otg_cart_" not in p]          # joint OTG (OTG_joints)
CART = [p for p in ALL if "otg_cart_" in p]             # Cartesian OTG (OTG_6dof_cartesian)
IDS = [os.path.basename(p)[:-4] for p in FIX]
CIDS = [os.path.basename(p)[:-4] for p in CART]


def test_fixture_set_is_complete():
    assert sorted(IDS) == sorted(G.scenarios()) and sorted(CIDS) == sorted(G.cart_scenarios())


@pytest.mark.parametrize("path", CART, ids=CIDS)
def test_cartesian_fixtures_are_reference_outputs(path):
    if otg_ref.build() is None:
        pytest.skip("oracle/_ref/libsaip_otg_ref.so not present (built only where /root/reference exists)")
    sc, steps, desired, reached, meta = G.load_cart(path)
    s2, d2, r2, moving = G.run_reference_cart(sc)
    assert np.array_equal(s2, steps) and np.array_equal(r2, reached) and moving == meta["finished_with_velocity"]
    assert np.array_equal(d2, desired)


@pytest.mark.parametrize("path", CART, ids=CIDS)
def test_cartesian_trajectory_invariants(path):
    sc, steps, desired, reached, meta = G.load_cart(path)
    R = desired[:, :, 3:12].reshape(desired.shape[0], -1, 3, 3)
    assert np.abs(R @ R.transpose(0, 1, 3, 2) - np.eye(3)).max() < 1e-12        # desired orientations are rotations
    lim = np.array(sc["limits"])
    changed = [p for s, k, p in sc["events"] if k == "limits"]
    amax_lin = max([lim[1]] + [p[1] for p in changed])
    amax_ang = max([lim[3]] + [p[3] for p in changed])
    # per-axis limits: each Cartesian / rotation-vector axis is its own Ruckig dof
    assert np.abs(desired[:, :, 18:21]).max() <= amax_lin + 1e-9
    ang_acc_ref = np.abs(desired[:, :, 21:24]).max()
    assert ang_acc_ref <= np.sqrt(3) * amax_ang + 1e-9                           # reported in the base frame: rotated per-axis values
    last = [p for s, k, p in sc["events"] if k == "goal"][-1]
    done = reached[-1]
    assert done.sum() >= done.size - 2
    assert np.abs(desired[-1][done, 0:3] - last[0][done]).max() < 1e-9
    assert np.abs(R[-1][done] - last[1][done]).max() < 1e-7


@pytest.mark.parametrize("path", FIX, ids=IDS)
def test_fixtures_are_reference_outputs(path):
    if otg_ref.build() is None:
        pytest.skip("oracle/_ref/libsaip_otg_ref.so not present (built only where /root/reference exists)")
    sc, steps, desired, reached, meta = G.load(path)
    s2, d2, r2, moving = G.run_reference(sc)
    assert np.array_equal(s2, steps) and np.array_equal(r2, reached) and moving == meta["finished_with_velocity"]
    assert np.array_equal(d2, desired)  # same machine code, same inputs: bit for bit


@pytest.mark.parametrize("path", FIX, ids=IDS)
def test_fixture_scenarios_match_generator(path):
    """the committed scenario data is what the seeded generator produces today"""
    sc, *_ = G.load(path)
    ref = G.scenarios()[os.path.basename(path)[:-4]]
    assert np.array_equal(sc["q0"], ref["q0"]) and sc["K"] == ref["K"] and len(sc["events"]) == len(ref["events"])
    for (s1, k1, p1), (s2, k2, p2) in zip(sc["events"], ref["events"]):
        assert s1 == s2 and k1 == k2
        if k1 == "reinit":
            assert np.array_equal(p1, p2)
        else:
            assert np.array_equal(p1[0], np.asarray(p2[0], float)) and np.array_equal(p1[1], np.asarray(p2[1], float))


@pytest.mark.parametrize("path", FIX, ids=IDS)
def test_trajectory_invariants(path):
    sc, steps, desired, reached, meta = G.load(path)
    pos, vel, acc = desired[:, 0], desired[:, 1], desired[:, 2]
    m = pos.shape[-1]
    vmax = np.broadcast_to(sc["vmax"], (m,)).copy()
    amax = np.broadcast_to(sc["amax"], (m,)).copy()
    lim_events = [(s, p) for s, k, p in sc["events"] if k == "limits"]
    for r, k in enumerate(steps):
        vm, am = vmax, amax
        for s, p in lim_events:
            if k >= s:
                am = np.broadcast_to(p[1], (m,))
        assert (np.abs(acc[r]) <= am + 1e-9).all()
        if not lim_events:
            assert (np.abs(vel[r]) <= vm + 1e-9).all()
    # acceleration-limited profiles are continuous in position and velocity: |dp| <= vmax_ever * dt * stride
    if not any(k == "reinit" for _, k, _ in sc["events"]):
        dstep = np.diff(steps)[:, None, None]
        assert (np.abs(np.diff(pos, axis=0)) <= np.abs(vel).max() * meta["dt"] * dstep * (1 + 1e-9) + 1e-12).all()
    # where the goal is reached the state sits on the last goal with zero velocity
    last_goal = [p for s, k, p in sc["events"] if k == "goal"][-1]
    done = reached[-1]
    assert done.sum() >= done.size - 1
    assert np.abs(pos[-1][done] - last_goal[0][done]).max() < 1e-9 and np.abs(vel[-1][done]).max() < 1e-3
