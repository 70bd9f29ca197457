"""soak: the same 32-chunk batch repeated; device memory in use after every repetition must stay flat
usage: python tools/soak.py [reps=6] [chunks=32] [logn=20]"""
import json, os, sys, tempfile, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
logn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = EngineConfig(air="chunk64", logn=logn, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_soak"), witness_threads=int(os.environ.get("ZP_SOAK_WT", "16")))
eng = Engine(default_backend_factory(0), cfg)
eng.groth16_keys()
first = None
for r in range(reps):
    t0 = time.perf_counter()
    ch = eng.gen_batch_chunks("b", list(range(1, K + 1)), 12345, "evm")
    proofs = eng.gen_chunk_proofs("b", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    agg = eng.aggregate("b", proofs[0]["proof"], proofs[-1]["proof"])
    fin = eng.final("b", agg, "BN128", "479881985774944702531460751064278034642760119942")
    dt = time.perf_counter() - t0
    dig = hashlib.sha256("".join(p["proof"] for p in proofs).encode()).hexdigest()[:16]
    first = first or dig
    info = eng.be.p.device_info()
    print(json.dumps({"rep": r, "wall_s": round(dt, 3), "used_GiB": round((info["total_mem"] - info["free_mem"]) / 2**30, 2),
                      "proofs_digest": dig, "same_as_first": dig == first}), flush=True)
