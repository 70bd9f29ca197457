"""end-to-end over gRPC on the GPU: serve() + the ProverChannel mirror of eigen-zeth's client, N blocks, metrics scrape
usage: python tools/service_e2e.py [blocks=2] [logn=20] [chunks_per_block=4]      (ZP_PREWARM=0: open the port at once, as rounds 1-4 did)"""
import json, os, sys, tempfile, time, urllib.request
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import EngineConfig
from eigen_zeth_amd.service.server import serve
from eigen_zeth_amd.service.client import ProverChannel
from eigen_zeth_amd.service.metrics import Metrics

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cpb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tmp = tempfile.mkdtemp()
cfg = EngineConfig(air="chunk64", logn=logn, chunks_per_block=cpb, crs_dir=os.path.join(tmp, "crs"), witness_threads=8)
prewarm = os.environ.get("ZP_PREWARM", "1") != "0"
t0 = time.perf_counter()
server, port = serve(0, "127.0.0.1", os.path.join(tmp, "state"), cfg, 0, metrics_port=0, prewarm=prewarm)
print(json.dumps({"service_start_s": round(time.perf_counter() - t0, 3), "prewarm": prewarm}), flush=True)
ch = ProverChannel("127.0.0.1:%d" % port)
for b in range(1, blocks + 1):
    t0 = time.perf_counter()
    res = ch.execute(b)
    dt = time.perf_counter() - t0
    print(json.dumps({"block": b, "wall_s": round(dt, 3), "proof_bytes": len(res["proof"]), "public_input": res["public_input"][:40],
                      "post_state_root": bytes(res["post_state_root"]).hex()[:16], "chunks": len(res["chunk_proofs"])}), flush=True)
ch.close()
server.stop(0)
print("OK")
