"""the final STARK in BN128-hash mode on one GPU (measurement tool): usage: python tools/final_stark_bench.py [air=chunk16] [logn=16] [logb=2] [queries=50]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR, prover as PR
from eigen_zeth_amd.stark.backend_hip import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "chunk16"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
logb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 50
air = AIR.get_air(name)
tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 3)
for mode in ("bn128", "gl"):
    be = HipBackend(hash_mode=mode)
    params = PR.StarkParams(logn, logb, 3, 5, nq, hash=mode)
    for rep in range(2):
        tm = {}
        t0 = time.perf_counter()
        proof = PR.prove(air, tr, pub, params, be, timings=tm)
        dt = time.perf_counter() - t0
    text = PR.proof_to_json(proof)
    for rep in range(2):       # the form the service runs: the whole STARK behind one C-ABI call (witness upload included)
        t0 = time.perf_counter()
        text_native = be.prove_native(air, tr, pub, params)
        dt_native = time.perf_counter() - t0
    print(json.dumps({"hash": mode, "air": name, "logn": logn, "logb": logb, "queries": nq, "wall_ms": round(dt * 1e3, 1),
                      "one_call_prover_wall_ms": round(dt_native * 1e3, 1), "same_proof_text": text_native == text,
                      "stages_ms": {k: round(v * 1e3, 2) for k, v in tm.items()}, "proof_bytes": len(text)}), flush=True)
