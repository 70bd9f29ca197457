"""Everything a COMPILED host needs to answer GenAggregatedProof for one shape of inner proofs, as files (the AIR and its witness schedule
are DATA: a Rust / C++ prover service needs neither Python nor a compiler at run time -- host/aggregate.cpp works from these):

  <out>/inner_program.bin     constraint program of the inner proofs' AIR (its SHA-256 is their "air_digest")
  <out>/verifier_program.bin  constraint program of the verifier AIR for this shape (what zp_stark_prove takes)
  <out>/witness_desc.bin      schedule of the verifier witness (what zp_recursion_witness takes)
  <out>/aggregate.txt         line 1: STARK parameters of the aggregation proof  logn logb fri_logf fri_final_log n_queries pow_bits
                              line 2: the text an aggregated proof starts with, with the batch id left as %s

usage: python tools/export_recursion_shape.py <out dir> [air=chunk64] [logn=20] [n_proofs=2] [n_queries=80] [pow_bits=20] [agg_queries=50]
(the other parameters are the service's defaults: blow-up 2, FRI fold 8, final layer 2^5)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from eigen_zeth_amd.poseidon_constants import default_mds, default_round_constants
from eigen_zeth_amd.service.engine import EngineConfig, aggregated_head
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from eigen_zeth_amd.stark import verifier_air as VA


def export(out, air_name="chunk64", logn=20, n_proofs=2, n_queries=80, pow_bits=20, agg_queries=50, cfg=None):
    cfg = cfg or EngineConfig(air=air_name, logn=logn, n_queries=n_queries, pow_bits=pow_bits, agg_queries=agg_queries)
    air = AIR.get_air(cfg.air)
    sp = PR.StarkParams(logn, cfg.logb, cfg.fri_logf, cfg.fri_final_log, cfg.n_queries, cfg.pow_bits)
    shape = VA.Shape(sp.logn, sp.logb, air.width, air.width2, 3 * AIR.quotient_chunks(air), sp.n_queries, sp.fri_logf, sp.fri_final_log, n_proofs,
                     air.n_pub, sp.pow_bits)
    vair = VA.verifier_air(shape, default_round_constants(), default_mds())
    ap = VA.aggregation_params(shape, cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log, cfg.agg_pow_bits)
    os.makedirs(out, exist_ok=True)
    np.asarray(air.program(), dtype=np.uint64).tofile(os.path.join(out, "inner_program.bin"))
    np.asarray(vair.program(), dtype=np.uint64).tofile(os.path.join(out, "verifier_program.bin"))
    np.asarray(VA.arith_descriptor(shape), dtype=np.uint64).tofile(os.path.join(out, "witness_desc.bin"))
    with open(os.path.join(out, "aggregate.txt"), "w") as f:
        f.write("%d %d %d %d %d %d\n" % (ap.logn, ap.logb, ap.fri_logf, ap.fri_final_log, ap.n_queries, ap.pow_bits))
        f.write(aggregated_head("%s", shape, 1, vair.digest()) + "\n")
    return shape, vair, ap


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a:
        raise SystemExit(__doc__)
    ints = [int(v) for v in a[2:]]
    shape, vair, ap = export(a[0], a[1] if len(a) > 1 else "chunk64", *ints)
    print("shape", shape.to_dict(), "-> verifier trace 2^%d rows, %d public inputs, program %d words" % (ap.logn, shape.n_pub(), len(vair.program())))
