"""Could the streaming kernel walk a CHAIN per workgroup (its 22 node blocks one after the other, 8.3 MB) instead of a node per
workgroup?  Then the leaf-to-top running sums of k_up_chain would ride in the streaming kernel's registers (one launch and ~11 us less
per iteration).  A do-nothing reader in both shapes (rn_measure_hbm_shape 0: n workgroups of 256 threads, each streaming its own
contiguous piece): 10 864 pieces of one node block vs 493 pieces of 22 blocks (+ 986 half-chains, 1 972 quarter-chains)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rapidnet_amd import capi, synth

p = synth.make_problem("tiny")
s = capi.Solver(p["network"], p["tree"], p["config"])
block = 376320
print("flat 2 GiB: read %.0f GB/s" % s.measureHbm(2 << 30, 3)[0])
for chunk, n, what in ((block, 10864, "one node block per workgroup (what k_stream_gemv does)"), (22 * block, 493, "one chain (22 blocks) per workgroup"),
                       (11 * block, 986, "half a chain per workgroup"), (block * 22 // 4, 1972, "a quarter chain per workgroup")):
    print("%-55s %9d B x %5d:" % (what, chunk, n), " ".join("unr%d %.0f" % (u, s.measureHbmShape(0, chunk, n, u, 3)) for u in (1, 2, 4, 8)), "GB/s")
s.close()
