"""The communicator set-up cannot hang a caller (failure detection; the reference exits on any failure,
Configuration.h:38-81): rank 0 of a TWO-rank ncclUniqueId whose peer never arrives gets RN_E_COMM back within the time-out, the
half-made communicator is aborted, and the context goes on working without one."""
import time

import numpy as np
import pytest

from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_comm_init_returns_within_the_timeout_when_no_peer_arrives():
    p = synth.make_problem("small")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    ref = capi.Solver(p["network"], p["tree"], p["config"], device=0)
    ref.initialiseSmpcController(dh, ah)
    want = ref.algorithmApg(20)
    want_u = ref.get(capi.BUF_U)
    ref.close()
    s = capi.Solver(p["network"], p["tree"], p["config"], device=0)
    uid = capi.comm_unique_id()
    t0 = time.time()
    with pytest.raises(capi.RapidNetError, match="time-out"):
        s.commInit(0, 2, uid, timeout=4.0)
    elapsed = time.time() - t0
    assert 3.5 < elapsed < 30.0, elapsed
    info = s.shardInfo()
    assert info["comm_ranks"] == 0 and info["nranks"] == 1, info      # no communicator, still an unsharded context
    s.commCheck()                                                      # nothing to report
    s.initialiseSmpcController(dh, ah)
    got = s.algorithmApg(20)
    assert np.array_equal(got, want) and np.array_equal(s.get(capi.BUF_U), want_u)
    # a second attempt is allowed (the first left nothing behind) and a one-rank communicator comes up at once
    t0 = time.time()
    s.commInit(0, 1, capi.comm_unique_id(), timeout=60.0)
    assert time.time() - t0 < 30.0 and s.shardInfo()["comm_ranks"] == 1
    s.commCheck()
    s.close()
