"""The host half of eigen_zeth_amd/csrc/r1cs.hip under AddressSanitizer + UBSan (built with -DZP_R1CS_HOST_ONLY by g++): the circuit-blob parser
with its ARITHMETIC TEMPLATES (round 6: rows over local wires, linear-combination pools, witness programs, instance tables), the host evaluator that
runs the witness programs and checks every row, and zp_wrap_assign over the assignment script and the openings record -- a valid stage B-2 case must
give the reference's public input, and seeded mutations of the blob, the script and the record must be evaluated or refused without a single
out-of-bounds access (tests/native/r1cs_fuzz.cpp)."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_circuit_blob_and_assignment_under_sanitizers(tmp_path, tables):
    from eigen_zeth_amd import native
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from eigen_zeth_amd.service import wrap_arith as WA, wrap_circuit as WC
    from eigen_zeth_amd.stark import air as AIR, prover as PR
    from cpu_wrap_backend import CpuWrapBackend
    bn = bn254_poseidon_params(17)
    cpu = CpuWrapBackend(*tables, hash_mode="bn128", bn_tables=bn)
    air = AIR.get_air("fib")
    tr, pub = native.synth_trace(air.trace_kind, 5, air.width, 5)
    params = PR.StarkParams(5, 2, 2, 3, 2, pow_bits=0, hash="bn128")        # the smallest stage B-2 circuit: 2 queries, two FRI layers
    proof = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu)))
    lay = WC.Layout.of_air(air, params)
    head = WC.head_values(air, params, proof["root32"], proof["shift"])
    wc = WC.wrap_circuit(lay, WA.Statement(air.program(), proof["root32"], proof["shift"], head))
    tlog = WC.TranscriptLog(proof, lay, head)
    rec = WC.openings_record(proof, lay, tlog)
    aux = [12345]
    set_idx, set_val = native.wrap_assign(wc.script, rec, aux)
    w = np.zeros((wc.c.n_wires, 4), dtype=np.uint64)
    mask = np.zeros(wc.c.n_wires, dtype=np.uint8)
    w[set_idx.astype(np.int64)] = set_val
    mask[set_idx.astype(np.int64)] = 1
    wf, _, _, _ = native.r1cs_eval(wc.blob, w, mask)
    u = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    case = np.concatenate([u([wc.blob.size]), u(wc.blob), u([set_idx.size]), u(set_idx), u(set_val), u(wf[1]), u([wc.script.size]), u(wc.script), u([rec.size]), u(rec),
                           u([len(aux)]), u(native.fr_words(aux))])
    path = str(tmp_path / "case.bin")
    case.tofile(path)
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17"]
    csrc = os.path.join(ROOT, "eigen_zeth_amd", "csrc")
    objs = []
    for src, extra in (("r1cs.hip", ["-DZP_R1CS_HOST_ONLY"]), ("verify.hip", [])):
        obj = str(tmp_path / (src + ".o"))
        subprocess.check_call(["g++", *san, "-x", "c++", "-D__HIP_PLATFORM_AMD__", *extra, "-I/opt/rocm/include", "-c", os.path.join(csrc, src), "-o", obj])
        objs.append(obj)
    exe = str(tmp_path / "r1cs_fuzz")
    subprocess.check_call(["g++", *san, os.path.join(ROOT, "tests", "native", "r1cs_fuzz.cpp"), *objs, "-o", exe, "-lpthread"])
    out = subprocess.run([exe, path, "600"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.startswith("ok:"), out.stdout
    print(out.stdout.strip())
