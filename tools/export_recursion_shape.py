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


def export_final(out, eng, n_proofs=2, logn=None):
    """Everything host/aggregate.cpp's `final` mode needs to answer GenFinalProof for aggregations of `n_proofs` chunk proofs of 2^logn rows,
    from an Engine with a GPU backend (the key's group elements are made on the GPU and written out: about 0.5 GB at the service's size):

      <out>/inner_program.bin, verifier_program.bin, witness_desc.bin   as above, one level up: the inner proof is the aggregation STARK,
                                                                        the verifier AIR over it is what the final STARK proves
      <out>/poseidon_bn254_t17.bin   [rp, rc words, mds words]: the tables zp_set_poseidon_bn254 takes
      <out>/wrap_circuit.bin, wrap_script.bin                            the R1CS of the wrap and its assignment script
      <out>/key_u1x.bin, key_v_wires.bin, key_v1x.bin, key_v2x.bin, key_l1.bin, key_h1.bin, key_delta1.bin   the proving key (points in the MSM layout)
      <out>/final.txt    line 1: logn logb fri_logf fri_final_log n_queries of the final STARK; line 2: the "circuit" text of the proof"""
    from eigen_zeth_amd import native
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from eigen_zeth_amd.service import groth16 as G16
    from eigen_zeth_amd.service import wrap_circuit as WC
    cfg, be = eng.cfg, eng.be
    air = AIR.get_air(cfg.air)
    sp = eng.stark_params(logn)
    chunk_shape = VA.Shape(sp.logn, sp.logb, air.width, air.width2, 3 * AIR.quotient_chunks(air), sp.n_queries, sp.fri_logf, sp.fri_final_log, n_proofs,
                           air.n_pub, sp.pow_bits, int(be.root32), int(be.shift))
    rc, mds = default_round_constants(), default_mds()
    agg_air = VA.verifier_air(chunk_shape, rc, mds)
    ap = VA.aggregation_params(chunk_shape, cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log, cfg.agg_pow_bits)
    agg_shape = VA.Shape(ap.logn, ap.logb, VA.WIDTH, 0, 3 * VA.Q_PIECES, ap.n_queries, ap.fri_logf, ap.fri_final_log, 1, chunk_shape.n_pub(), ap.pow_bits,
                         int(be.root32), int(be.shift))
    fair = VA.verifier_air(agg_shape, rc, mds)
    fp = VA.aggregation_params(agg_shape, cfg.final_queries, cfg.fri_logf, cfg.fri_final_log, 0, hash="bn128")
    wc, key = eng._wrap_key(fair, fp)
    os.makedirs(out, exist_ok=True)
    w = lambda name, arr, dt=np.uint64: np.ascontiguousarray(arr, dtype=dt).tofile(os.path.join(out, name))
    w("inner_program.bin", agg_air.program())
    w("verifier_program.bin", fair.program())
    w("witness_desc.bin", VA.arith_descriptor(agg_shape))
    brc, bmds, rp = bn254_poseidon_params(17)
    w("poseidon_bn254_t17.bin", np.concatenate([np.array([rp], dtype=np.uint64), native.fr_words(brc).reshape(-1), native.fr_words([v for row in bmds for v in row]).reshape(-1)]))
    w("wrap_circuit.bin", wc.blob)
    w("wrap_script.bin", wc.script)
    dev = key.load_points(be)
    for name in ("u1x", "v1x", "v2x", "l1", "h1"):
        d, n = dev[name]
        w("key_%s.bin" % name, be.p.download(d, (n * (32 if name == "v2x" else 16) // 2,)))
    w("key_v_wires.bin", be.p.download(dev["v_wires"][0], ((dev["v_wires"][1] + 1) // 2,)).view(np.uint32)[:dev["v_wires"][1]], np.uint32)
    w("key_delta1.bin", G16._g1_words(dev["delta1"]), np.uint32)
    with open(os.path.join(out, "final.txt"), "w") as f:
        f.write("%d %d %d %d %d\n" % (fp.logn, fp.logb, fp.fri_logf, fp.fri_final_log, fp.n_queries))
        f.write(G16.circuit_text(wc, key) + "\n")
    return wc, key, fp


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a:
        raise SystemExit(__doc__)
    ints = [int(v) for v in a[2:]]
    shape, vair, ap = export(a[0], a[1] if len(a) > 1 else "chunk64", *ints)
    print("shape", shape.to_dict(), "-> verifier trace 2^%d rows, %d public inputs, program %d words" % (ap.logn, shape.n_pub(), len(vair.program())))
