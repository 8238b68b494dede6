"""Step-by-step timing of the communicator time-out path on one GPU (every step is logged when it starts and when it ends)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rapidnet_amd import capi, synth

T0 = time.time()


def say(msg):
    print("[%6.1f s] %s" % (time.time() - T0, msg), flush=True)


p = synth.make_problem("small")
dh, ah = synth.forecast_at(p["forecast"], 0)
s = capi.Solver(p["network"], p["tree"], p["config"], device=0)
say("context created")
uid = capi.comm_unique_id()
say("unique id")
try:
    s.commInit(0, 2, uid, timeout=4.0)
    say("commInit(0, 2) returned WITHOUT an error?!")
except capi.RapidNetError as e:
    say("commInit(0, 2) raised: %s" % e)
say("shardInfo: %s" % s.shardInfo())
s.commCheck()
say("commCheck done")
s.initialiseSmpcController(dh, ah)
h = s.algorithmApg(20)
say("20 iterations: last primal infeasibility %g" % h[-1])
uid2 = capi.comm_unique_id()
say("second unique id")
try:
    s.commInit(0, 1, uid2, timeout=30.0)
    say("commInit(0, 1) ok: %s" % s.shardInfo())
except capi.RapidNetError as e:
    say("commInit(0, 1) raised: %s" % e)
s.close()
say("closed")
os._exit(0)
