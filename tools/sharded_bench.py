"""What sharding ONE chunk proof costs in orchestration (measurement tool): zp_stark_prove on one ctx against zp_stark_prove_sharded
with G thread-ranks sharing the SAME GPU (in-process communicator: collectives are device copies + a thread barrier).  On one GPU the
ranks compete for the same CUs, so G ranks cannot be faster than one; the difference to the single-ctx time is the price of the
extra exchange / gather steps and of G host threads walking the protocol -- the part that does not shrink with more GPUs.
usage: python tools/sharded_bench.py [logn=20] [air=chunk64] [bn128]   (bn128: zp_stark_prove_bn128 against zp_stark_prove_sharded_bn128)"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
air = AIR.get_air(sys.argv[2] if len(sys.argv) > 2 else "chunk64")
tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 21)
pub = [int(v) for v in pub]
bn = len(sys.argv) > 3 and sys.argv[3] == "bn128"
args = (logn, 1, 4, 5, 64) if bn else (logn, 1, 3, 5, 80, 20)
p0 = native.Prover(0)
if bn:
    p0.install_poseidon_bn254(17)
d = p0.upload(tr)
for _ in range(3):
    t0 = time.perf_counter()
    single = p0.stark_prove_bn128(air.name, air.program(), d, pub, *args) if bn else p0.stark_prove(air.name, air.program(), d, pub, *args)
    t_single = time.perf_counter() - t0
d.free()
print(json.dumps({"air": air.name, "logn": logn, "ranks": 1, "entry": "zp_stark_prove_bn128" if bn else "zp_stark_prove", "wall_ms": round(t_single * 1e3, 2)}), flush=True)
for G in (2, 4, 8):
    wl = -(-air.width // G)
    group = native.CommGroup(G)
    walls, same = [0.0] * G, [False] * G
    start = threading.Barrier(G)

    def body(r):
        p = native.Prover(0)
        c = native.Comm(p, r, G, group=group)
        d_l = p.upload(np.ascontiguousarray(tr[r * wl:(r + 1) * wl]))
        for _ in range(3):
            start.wait()
            t0 = time.perf_counter()
            text = c.stark_prove_sharded(air.name, air.program(), d_l, pub, *args, bn128=bn)
            walls[r] = time.perf_counter() - t0
        same[r] = text == single
        d_l.free()
        c.close()
        p.close()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    group.close()
    print(json.dumps({"air": air.name, "logn": logn, "ranks": G, "entry": "zp_stark_prove_sharded%s (thread-ranks on one GPU)" % ("_bn128" if bn else ""),
                      "wall_ms_max_over_ranks": round(max(walls) * 1e3, 2), "same_proof_text": all(same)}), flush=True)
