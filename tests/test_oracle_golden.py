"""Pins the CPU oracle (oracle/orl_oracle.c) to the reference: every golden trace captured by
importing the reference (oracle/gen_golden.py) must be reproduced bit-for-bit — integers with ==,
float64 values with == as well (the oracle uses the same libm and the same operation order)."""
import numpy as np
import pytest

from oracle.oracle import OracleBatch
from tests.helpers import golden_names, load_golden, replay, replay_h, replay_q, replay_w


def _make(meta):
    kw = dict(meta["kwargs"])
    seed = kw.pop("seed")
    return OracleBatch(meta["env"], meta["topology"], [seed], **kw)


def _exact(name):
    def check(t, what, got, exp):
        if got is None:
            return
        got, exp = np.asarray(got), np.asarray(exp)
        if got.dtype.kind == "f" or exp.dtype.kind == "f":
            ok = np.array_equal(got.astype(np.float64), exp.astype(np.float64), equal_nan=True)
        else:
            ok = np.array_equal(got, exp)
        assert ok, "%s: step %d: %s differs\n got %r\n exp %r" % (name, t, what, got, exp)
    return check


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_reference_trace(name):
    g = load_golden(name)
    env = _make(g["meta"])
    replay(env, g, _exact(name))


@pytest.mark.parametrize("name", golden_names("w"))
def test_oracle_reproduces_wrapper_and_event_fixtures(name):
    """PathOnlyFirstFitAction, SimpleMatrixObservation, the 2-D action histograms, seed() and reset(full) mid-run, as
    captured from the reference by oracle/gen_golden_wrappers.py."""
    g = load_golden(name)
    env = _make(g["meta"])
    replay_w(env, g, _exact(name))


def test_oracle_reproduces_rmcsa_4d_action_histograms():
    """RMCSAEnv.actions_output / actions_taken (rmcsa_env.py:145-180, 219, 273, 284-289; cleared by a full reset, :437-454)."""
    g = load_golden("h1_rmcsa_hist4d")
    replay_h(_make(g["meta"]), g, _exact("h1_rmcsa_hist4d"))


@pytest.mark.parametrize("name", golden_names("q"))
def test_oracle_reproduces_qos_fixtures(name):
    """QoSConstrainedRA (qos_constrained_ra.py) as captured from the reference with its constructor repaired at import
    time (oracle/gen_golden_qos.py): three heuristics and a stored action stream, three service classes."""
    g = load_golden(name)
    env = _make(g["meta"])
    replay_q(env, g, _exact(name))
