"""single-chunk full STARK on one GPU with per-stage wall-clock (measurement tool)
usage: python tools/stark_bench.py [air] [logn] [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR, prover as PR
from eigen_zeth_amd.stark.backend_hip import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "wide64"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
air = AIR.get_air(name)
t0 = time.perf_counter(); tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 42); tw = time.perf_counter() - t0
be = HipBackend(0)
if os.environ.get('ZP_COOP_LOG'): be.p.set_tuning('merkle_coop_log', int(os.environ['ZP_COOP_LOG']))
params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=5, n_queries=80, pow_bits=20)   # the service default (100 bits)
for r in range(reps):
    tm = {}
    proof = PR.prove(air, tr, pub, params, be, timings=tm)
    print(json.dumps({"air": name, "logn": logn, "rep": r, "witness_s": round(tw, 3), "proof_bytes": len(PR.proof_to_json(proof)),
                      "stages_ms": {k: round(v * 1e3, 2) for k, v in tm.items()}}), flush=True)
if os.environ.get("ZP_NATIVE", "1") != "0":      # the same proof through zp_stark_prove (one C-ABI call)
    for r in range(reps):
        t0 = time.perf_counter()
        text = be.prove_native(air, tr, pub, params)
        dt = time.perf_counter() - t0
        print(json.dumps({"air": name, "logn": logn, "rep": r, "native_one_call_ms": round(dt * 1e3, 2), "proof_bytes": len(text),
                          "same_text": text == PR.proof_to_json(proof)}), flush=True)
