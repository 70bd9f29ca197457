"""Batch-proof wall-clock on one GPU (BASELINE configs[4] shape at N=1): K blocks -> K chunk STARKs ->
aggregate -> Groth16 wrap, through the engine (no gRPC).  usage: python tools/batch_bench.py [K=16] [logn=20] [air=chunk64]"""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('ZP_PRE_TORCH'):      # experiment: does a torch CUDA context in the process change the batch time?
    import torch
    torch.cuda.init(); _t = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize()
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
air = sys.argv[3] if len(sys.argv) > 3 else "chunk64"
cfg = EngineConfig(air=air, logn=logn, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs"), witness_threads=16, prover_streams=int(os.environ.get('ZP_STREAMS', '8')))
eng = Engine(default_backend_factory(0), cfg)
eng.pregenerate_witnesses = bool(os.environ.get("ZP_PREGEN"))
eng.be
t0 = time.perf_counter(); eng.groth16_keys(); t_crs = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter()
    ch = eng.gen_batch_chunks("b", list(range(1, K + 1)), 12345, "evm")
    if eng.pregenerate_witnesses:
        eng.prepare_witnesses(ch["batch_data"])
        t0 = time.perf_counter()
    t1 = time.perf_counter()
    proofs = eng.gen_chunk_proofs("b", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    t2 = time.perf_counter()
    agg = eng.aggregate("b", proofs[0]["proof"], proofs[-1]["proof"])
    fin = eng.final("b", agg, "BN128", "479881985774944702531460751064278034642760119942")
    t3 = time.perf_counter()
    stages = {}
    for k, v in eng.stage_timings.items():
        if "/" in k and not k.startswith("final"):
            for kk, vv in v.items():
                stages[kk] = stages.get(kk, 0.0) + vv
    if rep == 1 and os.environ.get("ZP_BATCH_VERBOSE"):
        for k, v in eng.stage_timings.items():
            print(k, {kk: round(vv * 1e3, 1) for kk, vv in v.items()}, file=sys.stderr)
    print(json.dumps({"rep": rep, "chunks": K, "logn": logn, "air": air, "batch_wall_s": t3 - t0, "chunks_s": t1 - t0,
                      "chunk_proofs_s": t2 - t1, "aggregate_final_s": t3 - t2, "crs_setup_s(one-time)": t_crs,
                      "sum_stage_s": {k: round(v, 3) for k, v in stages.items()},
                      "proof_bytes_total": sum(len(p["proof"]) for p in proofs)}), flush=True)
