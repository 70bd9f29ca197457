"""same-box A/B of the small tree levels (<= 2^15 nodes: merkle_subtree_kernel): the permutation state exchanged through LDS with two workgroup
barriers per round (rounds 1-4) against 64-bit wave shuffles with one barrier per level (knob merkle_top_wave = 1, round 5).  Trees of 4-value
leaves (no leaf hashing: the tree levels alone), and one chunk proof.  Measurement tool.  usage: python tools/merkle_top_ab.py > profiles/r5_merkle_top_ab.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd import native
from eigen_zeth_amd.native import Prover

p = Prover(0)
print("# tools/merkle_top_ab.py on one MI355X: ms per tree (median of 15, alternating), leaves of 4 values (the levels alone)")
for logm in (7, 10, 13, 15, 18, 20):
    M = 1 << logm
    cols = np.random.default_rng(logm).integers(0, 2**62, size=(4, M), dtype=np.uint64)
    d, tree = p.upload(cols), p.alloc((2 * M - 1) * 4)
    res = {0: [], 1: []}
    for rep in range(16):
        for knob in (0, 1):
            p.set_tuning("merkle_top_wave", knob)
            p.sync()
            t0 = time.perf_counter()
            p.merkle_commit(d, M, 4, tree)
            p.sync()
            if rep:
                res[knob].append((time.perf_counter() - t0) * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print("2^%d leaves: LDS + barriers %.3f ms   wave shuffles %.3f ms   (%.2fx)" % (logm, med[0], med[1], med[0] / med[1]), flush=True)
    d.free(); tree.free()
from eigen_zeth_amd.stark import air as AIR, prover as PR
from eigen_zeth_amd.stark.backend_hip import HipBackend
air = AIR.get_air("chunk64")
for logn in (20, 22):
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 42)
    be = HipBackend(prover=p)
    params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=5, n_queries=80, pow_bits=20)
    d_tr = p.upload(tr)
    texts, res = {}, {0: [], 1: []}
    for rep in range(6):
        for knob in (0, 1):
            p.set_tuning("merkle_top_wave", knob)
            p.sync()
            t0 = time.perf_counter()
            texts[knob] = p.stark_prove(air.name, air.program(), d_tr, [int(v) for v in pub], params.logn, params.logb, params.fri_logf, params.fri_final_log,
                                        params.n_queries, params.pow_bits)
            if rep:
                res[knob].append((time.perf_counter() - t0) * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print("chunk proof 2^%d rows x 76 columns (zp_stark_prove, trace resident): LDS form %.2f ms   wave shuffles %.2f ms   identical text: %s"
          % (logn, med[0], med[1], texts[0] == texts[1]), flush=True)
    d_tr.free()
