"""eigen_zeth_amd/csrc/proofparse.hip under AddressSanitizer + UBSan on the host (the GPU pool has no sanitizer runs): the parser of the
client's recursive-proof text (proto/prover/v1/prover.proto:115-148) refuses hostile documents -- repeated keys, FRI layers beyond the
ones sized, sizes that disagree with the scan -- instead of writing through pointers nobody set (round-4 advisor finding)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_proof_text_parser_under_sanitizers(tmp_path):
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-std=c++17"]
    obj, exe = str(tmp_path / "proofparse.o"), str(tmp_path / "proofparse_fuzz")
    # the file is host code: it builds as plain C++ (the HIP headers it sees through ctx.hpp only need the platform macro)
    subprocess.check_call(["g++", *san, "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c",
                           os.path.join(ROOT, "eigen_zeth_amd", "csrc", "proofparse.hip"), "-o", obj])
    subprocess.check_call(["g++", *san, os.path.join(ROOT, "tests", "native", "proofparse_fuzz.cpp"), obj, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:") and " 0 accepted by the scan and refused by the write pass" in out.stdout, out.stdout
