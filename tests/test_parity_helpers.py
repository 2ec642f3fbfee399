"""CPU tests of the parity helpers in conftest.py themselves: the GPU tests lean on them, so their logic is exercised here
on small synthetic conf matrices (oracle.mutual_nn_match restates GATs_SuperGlue.py:220-237)."""
import numpy as np
import pytest

from conftest import check_matches_outside_flips
from oracle import gatsspg_oracle as orc


def _case(seed=0, b=2, n1=12, n2=17):
    rs = np.random.RandomState(seed)
    conf = rs.uniform(0.0, 1.0, (b, n1, n2)).astype(np.float32)
    for i in range(6):                       # a few mutual nearest neighbours above the threshold
        conf[0, i, 2 * i] = 3.0 + i
    return conf


def _golden(conf, thr):
    m = orc.mutual_nn_match(conf, thr)
    return {"indices0_raw": m["indices0_raw"].astype(np.int32), "indices1_raw": m["indices1_raw"].astype(np.int32),
            "matches0": m["matches0"][0], "matches1": m["matches1"][0]}


def _pred0(conf, thr):
    m = orc.mutual_nn_match(conf, thr)
    return {"matches0": m["matches0"][0], "matches1": m["matches1"][0]}


def test_a_near_tie_swap_touches_only_what_it_can_reach():
    ref = _case()
    ref[0, 7, 3] = 2.0
    ref[0, 7, 9] = 2.0 * (1 - 1e-6)          # the reference's near-tie in row 7: columns 3 (winner) and 9
    ours = ref.copy()
    ours[0, 7, 3], ours[0, 7, 9] = ref[0, 7, 9], ref[0, 7, 3]      # our arithmetic resolves it the other way
    g = _golden(ref, 0.5)
    n = check_matches_outside_flips(ours, _pred0(ours, 0.5), g, "swap", 2e-5)
    assert n >= 1
    # untouched rows keep their matches (checked inside); the planted mutual matches are among them
    assert (_pred0(ours, 0.5)["matches0"][:6] == g["matches0"][:6]).all()


def test_a_wrong_match_elsewhere_is_caught():
    ref = _case(1)
    ref[0, 7, 3] = 2.0
    ref[0, 7, 9] = 2.0 * (1 - 1e-6)
    ours = ref.copy()
    ours[0, 7, 3], ours[0, 7, 9] = ref[0, 7, 9], ref[0, 7, 3]
    g = _golden(ref, 0.5)
    pred = _pred0(ours, 0.5)
    pred["matches0"] = pred["matches0"].copy()
    pred["matches0"][2] = -1 if pred["matches0"][2] >= 0 else 5    # a difference no flip explains
    with pytest.raises(AssertionError, match="no arg-max flip touches"):
        check_matches_outside_flips(ours, pred, g, "broken", 2e-5)


def test_a_flip_that_is_not_a_top2_swap_is_caught():
    ref = _case(2)
    ours = ref.copy()
    ours[1, 4, :] = ref[1, 4, ::-1]          # sample 1, row 4: a different winner that is no near-tie of anything
    g = _golden(ref, 0.5)
    if (ours.argmax(2) == ref.argmax(2)).all():
        pytest.skip("the reversal kept the arg-max")
    with pytest.raises(AssertionError):
        check_matches_outside_flips(ours, _pred0(ours, 0.5), g, "not-a-swap", 2e-5)
