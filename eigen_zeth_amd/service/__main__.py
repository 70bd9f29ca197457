"""python -m eigen_zeth_amd.service [--port 50061] [--state-dir DIR] [--air chunk64] [--logn 20] [--no-prewarm]"""
import argparse
import time

from .engine import EngineConfig
from .server import serve


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=50061)   # PROVER_ADDR default, src/config/env.rs:21
    ap.add_argument("--state-dir", default="prover_state")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--air", default="chunk64")
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--chunks-per-block", type=int, default=1)
    ap.add_argument("--logb", type=int, default=1, help="log2 of the LDE blow-up (1 bit of soundness per query and unit)")
    ap.add_argument("--n-queries", type=int, default=80, help="FRI queries per chunk proof")
    ap.add_argument("--pow-bits", type=int, default=20, help="proof-of-work grinding bits before the query phase")
    ap.add_argument("--l2-addr", default=None, help="L2 JSON-RPC (ZETH_L2_ADDR) to fetch block inputs from")
    ap.add_argument("--devices", default=None, help="comma-separated GPU ids to spread chunk proofs over (default: --device)")
    ap.add_argument("--metrics-port", type=int, default=None, help="serve Prometheus text metrics on /metrics")
    ap.add_argument("--agg-queries", type=int, default=50, help="queries of the aggregation STARK over the Merkle-verifier AIR (blow-up 4: 2 bits each)")
    ap.add_argument("--final-queries", type=int, default=50, help="queries of the final STARK (BN128-hash mode, Merkle-verifier AIR over the aggregated proof; blow-up 4, no grinding)")
    ap.add_argument("--aggregate-all-chunks", action="store_true",
                    help="GenAggregatedProof verifies EVERY chunk proof of the batch when the request names its first and last one (default: the two named proofs, as the wire contract says)")
    ap.add_argument("--no-prewarm", action="store_true",
                    help="open the port at once; the first request then also pays for the wrap key, the transform plans and tables, the kernels (seconds). "
                         "Default: one synthetic batch is proven end to end before the port opens")
    ap.add_argument("--final-ranks", type=int, default=1,
                    help="GenFinalProof's STARK as ONE proof over this many ranks of the process (a power of two; zp_stark_prove_sharded_bn128 on an in-process "
                         "communicator), rank r on the r-th id of --final-devices (default: all on --device, which only rehearses the path)")
    ap.add_argument("--final-devices", default=None, help="comma-separated GPU ids of the ranks of --final-ranks")
    a = ap.parse_args()
    if a.final_ranks < 1 or a.final_ranks > 64 or a.final_ranks & (a.final_ranks - 1):
        ap.error("--final-ranks must be a power of two between 1 and 64")
    if a.final_devices and len(a.final_devices.split(",")) != a.final_ranks:
        ap.error("--final-devices must name one GPU id per rank of --final-ranks")
    server, port = serve(a.port, a.host, a.state_dir, EngineConfig(a.air, a.logn, logb=a.logb, chunks_per_block=a.chunks_per_block, l2_addr=a.l2_addr, n_queries=a.n_queries, pow_bits=a.pow_bits,
                                                               agg_queries=a.agg_queries, final_queries=a.final_queries, aggregate_all_chunks=a.aggregate_all_chunks, final_ranks=a.final_ranks,
                                                               final_devices=[int(x) for x in a.final_devices.split(',')] if a.final_devices else None), a.device,
                         metrics_port=a.metrics_port,
                         devices=[int(x) for x in a.devices.split(',')] if a.devices else None, prewarm=not a.no_prewarm)
    print("prover.v1.ProverService listening on %s:%d  (chunk STARKs: %d queries x blow-up %d + %d grinding bits = %d bits conjectured)"
          % (a.host, port, a.n_queries, 1 << a.logb, a.pow_bits, a.n_queries * a.logb + a.pow_bits), flush=True)
    try:
        while True:
            time.sleep(3600)
    except KeyboardInterrupt:
        server.stop(1)


if __name__ == "__main__":
    main()
