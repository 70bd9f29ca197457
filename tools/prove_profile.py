"""cProfile of one chunk proof on the GPU backend (where does the host time go)"""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR, prover as PR
from eigen_zeth_amd.stark.backend_hip import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "chunk64"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
air = AIR.get_air(name)
tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 42)
be = HipBackend(0)
params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=5, n_queries=80, pow_bits=20)   # the service default (100 bits)
for _ in range(2):
    PR.prove(air, tr, pub, params, be)
pr = cProfile.Profile()
pr.enable()
proof = PR.prove(air, tr, pub, params, be)
js = PR.proof_to_json(proof)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
