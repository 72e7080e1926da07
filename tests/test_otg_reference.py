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
    ia = 3 if len(lim) == 4 else 4                                               # jerk-limited fixtures: (v, a, j) linear, (v, a, j) angular
    amax_lin = max([lim[1]] + [p[1] for p in changed])
    amax_ang = max([lim[ia]] + [p[ia] for p in changed])
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
        elif k1 == "mode":
            assert p1[0] == p2[0] and np.array_equal(p1[1], p2[1]) and np.allclose(p1[2], p2[2], rtol=0, atol=0)
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
                # (jerk-limited mode: an acceleration above a newly lowered limit comes down at the jerk limit -- the brake
                # pre-trajectory -- so only the old bound holds right away)
                am = np.broadcast_to(p[1], (m,)) if len(p) == 2 else np.maximum(amax, np.broadcast_to(p[1], (m,)))
        assert (np.abs(acc[r]) <= am + 1e-9).all()
        if not lim_events:
            assert (np.abs(vel[r]) <= vm + 1e-9).all()
    # acceleration-limited profiles are continuous in position and velocity: |dp| <= vmax_ever * dt * stride
    if not any(k in ("reinit", "mode") for _, k, _ in sc["events"]):   # (a re-initialisation, also the one of a mode switch, jumps)
        dstep = np.diff(steps)[:, None, None]
        assert (np.abs(np.diff(pos, axis=0)) <= np.abs(vel).max() * meta["dt"] * dstep * (1 + 1e-9) + 1e-12).all()
    # where the goal is reached the state sits on the last goal with zero velocity
    last_goal = [p for s, k, p in sc["events"] if k == "goal"][-1]
    done = reached[-1]
    assert done.sum() >= done.size - 1
    assert np.abs(pos[-1][done] - last_goal[0][done]).max() < 1e-9 and np.abs(vel[-1][done]).max() < 1e-3


@pytest.mark.parametrize("path", [p for p in FIX if "jerk" in p], ids=[i for i in IDS if "jerk" in i])
def test_jerk_limited_trajectories_respect_the_jerk_limit(path):
    """third-order profiles: the acceleration is continuous and changes by at most max_jerk * dt per control cycle (while the jerk-limited
    mode is on and away from re-initialisations, which reset the state)"""
    sc, steps, desired, reached, meta = G.load(path)
    acc = desired[:, 2]
    m = acc.shape[-1]
    jm = None if sc["jmax"] is None else np.broadcast_to(sc["jmax"], (m,)).copy()
    resets = [s for s, k, _ in sc["events"] if k in ("mode", "reinit")]
    checked = 0
    for r in range(1, len(steps)):
        if steps[r] != steps[r - 1] + 1:
            continue
        k = int(steps[r])
        jcur = jm
        for s, kind, p in sc["events"]:
            if s <= k and kind == "limits" and len(p) == 3:
                jcur = np.maximum(jcur, np.broadcast_to(p[2], (m,))) if jcur is not None else np.broadcast_to(p[2], (m,))
            if s <= k and kind == "mode":
                jcur = np.broadcast_to(p[2][2], (m,)) if p[0] == "jerk" else None
        if jcur is None or any(s - 1 <= k <= s + 1 for s in resets):
            continue
        assert (np.abs(acc[r] - acc[r - 1]) <= jcur * meta["dt"] * (1 + 1e-9) + 1e-9).all(), k
        checked += 1
    assert checked > 5


def test_jerk_limited_core_matches_the_reference_classes():
    """function level: the product's third-order solver (csrc/saip_otg3.h compiled for the host) against the reference's
    PositionThirdOrderStep1 / Step2 / BrakeProfile on 60 000 random inputs (oracle/_ref/otg3_compare, built by `make -C oracle ref`)"""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "otg3_compare")
    if otg_ref.build() is None or not os.path.exists(exe):
        pytest.skip("oracle/_ref/otg3_compare not present (built only where /root/reference exists)")
    r = subprocess.run([exe, "60000"], capture_output=True, text=True, timeout=300)
    print(r.stdout[-400:])
    assert r.returncode == 0 and "mismatches 0" in r.stdout
