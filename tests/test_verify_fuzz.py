"""eigen_zeth_amd/csrc/verify.hip (zp_program_eval_ext + the program-table parser) under AddressSanitizer + UBSan on the host: the verifier's
side of a constraint program is fed by a client's proof text (public inputs, evaluations) and takes a blob through a C ABI -- a valid case of a
statement with sparse periodic fixed columns (the chunk-level verifier AIR of a small shape) must give the checker's values, and thousands of
seeded mutations of the blob, the public inputs and the evaluations must be evaluated or refused without a single out-of-bounds access."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_program_evaluation_under_sanitizers(tmp_path, tables):
    from eigen_zeth_amd import native
    from eigen_zeth_amd.stark import air as AIR, prover as PR, verifier_air as VA
    from oracle import oracle as O
    from oracle.stark_cpu import CpuBackend
    rc, mds = tables
    cpu = CpuBackend(rc, mds)
    air = AIR.get_air("fib")
    params = PR.StarkParams(5, 1, 2, 3, 3, pow_bits=0)
    tr, pub = native.synth_trace(air.trace_kind, 5, air.width, 9)
    proof = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu)))
    shape = VA.Shape.of_proof(proof, 1)
    vair = VA.verifier_air(shape, rc, mds)                       # 47 columns, ~100 sparse fixed columns, public-input entries
    prog = np.ascontiguousarray(vair.program(), dtype=np.uint64)
    _, pubs = VA.build_witness(shape, [proof], cpu, air.digest_words())
    logn = shape.logn_trace()
    Wt = vair.width + vair.width2
    ev_z, ev_zw, zeta = O.random_field((Wt, 3), 71), O.random_field((Wt, 3), 72), O.random_field((3,), 73)
    want = native.program_eval_ext(prog, [int(v) for v in pubs], logn, O.ROOT32_DEFAULT, zeta, ev_z, ev_zw)
    case = np.concatenate([np.array([prog.size], dtype=np.uint64), prog, np.array([len(pubs)], dtype=np.uint64), np.asarray(pubs, dtype=np.uint64),
                           np.array([logn, O.ROOT32_DEFAULT], dtype=np.uint64), zeta.astype(np.uint64), ev_z.reshape(-1), ev_zw.reshape(-1), want.reshape(-1)])
    path = str(tmp_path / "case.bin")
    case.tofile(path)
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17"]
    obj, exe = str(tmp_path / "verify.o"), str(tmp_path / "verify_fuzz")
    subprocess.check_call(["g++", *san, "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c",
                           os.path.join(ROOT, "eigen_zeth_amd", "csrc", "verify.hip"), "-o", obj])
    subprocess.check_call(["g++", *san, os.path.join(ROOT, "tests", "native", "verify_fuzz.cpp"), obj, "-o", exe, "-lpthread"])
    out = subprocess.run([exe, path, "2500"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:"), out.stdout
