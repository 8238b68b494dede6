"""HBM read-ceiling probes on the GPU box: flat grid-stride reader vs readers in the streaming kernel's shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rapidnet_amd import capi, synth

p = synth.make_problem("tiny")
s = capi.Solver(p["network"], p["tree"], p["config"])
print("lib", capi.lib_path())
print("flat 2 GiB: read %.0f GB/s copy %.0f GB/s" % s.measureHbm(2 << 30, 3))
for chunk, n in ((376320, 10864), (1 << 20, 4096), (94080, 43456)):
    print("chunk per workgroup, %8d B x %6d:" % (chunk, n), " ".join("unr%d %.0f" % (u, s.measureHbmShape(0, chunk, n, u, 3)) for u in (1, 2, 4, 8)))
if "--lockstep" in sys.argv:
    total = 4 << 30
    for piece in (7840, 8192, 4096):
        for n in (512, 768, 1024, 2048):
            print("lockstep pieces of %5d B, %5d workgroups:" % (piece, n), " ".join("unr%d %.0f" % (u, s.measureHbmShape(1, piece, n, u, 3, total)) for u in (1, 2, 4, 8)))
