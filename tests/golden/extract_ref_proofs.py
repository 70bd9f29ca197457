"""Collect every proof / public-input vector the reference holds, with the expected outputs its own tests assert.

Run HERE (where /root/reference exists):  python tests/golden/extract_ref_proofs.py  -> tests/golden/ref_proofs.json

The reference has exactly one test that pins a RESULT on proof data: `test_parse_proof` (src/settlement/ethereum/mod.rs:487-571) feeds a
snarkjs-style JSON text to `parse_proof` (:445-474) and asserts the eight U256 it returns, field by field, in the order a.x, a.y, b.x[0],
b.x[1], b.y[0], b.y[1], c.x, c.y; `test_parse_public_input` (:573-589) does the same for the public input.  Two more proofs appear as
inputs of ignored integration tests (src/settlement/worker.rs:760-761 -- the same text again at src/settlement/custom/methods.rs:710-711)
and one as the debug stand-in file (proof/proof.json, proof/public_input.json).  The output is DATA: the JSON inputs as parsed values and
the asserted outputs as decimal strings, each with its source location; no source text of the reference is stored."""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_proofs.json")


def rust_string(lit):
    """the value of a Rust "..." literal body (escapes \\" \\n \\\\ only)"""
    return lit.replace('\\"', '"').replace("\\n", "\n").replace("\\\\", "\\")


def main():
    eth = open(os.path.join(REF, "src/settlement/ethereum/mod.rs")).read()
    lines = eth.split("\n")

    def line_of(pos):
        return eth.count("\n", 0, pos) + 1

    fixtures = []
    # 1. test_parse_proof: raw-string input + asserted fields
    t0 = eth.index("fn test_parse_proof()")
    t1 = eth.index("fn test_parse_public_input()")
    body = eth[t0:t1]
    m = re.search(r'r#"(.*?)"#', body, re.S)
    proof = json.loads(m.group(1))
    asserted = re.findall(r"assert_eq!\(\s*proof\.([abc])\.([xy])(?:\[(\d)\])?,\s*U256::from_dec_str\(\s*\"(\d+)\"", body)
    order = ["%s.%s%s" % (pt, co, "[%s]" % ix if ix else "") for pt, co, ix, _ in asserted]
    fixtures.append({
        "source": "src/settlement/ethereum/mod.rs:%d-%d (input), :%d-%d (asserted outputs)" % (
            line_of(t0 + m.start()), line_of(t0 + m.end()), line_of(t0 + body.index("assert_eq!")), line_of(t1) - 3),
        "pinned_by_reference_test": "test_parse_proof",
        "proof": proof,
        "expected_fields_in_order": order,
        "expected_u256_in_order": [v for _, _, _, v in asserted],
    })
    # 2. test_parse_public_input
    t2 = eth.index("fn test_from_conf_path()", t1)
    body = eth[t1:t2]
    m = re.search(r'r#"(.*?)"#', body, re.S)
    want = re.search(r"input\[0\],\s*U256::from_dec_str\(\s*\"(\d+)\"", body).group(1)
    public_inputs = [{
        "source": "src/settlement/ethereum/mod.rs:%d-%d" % (line_of(t1), line_of(t2) - 3),
        "pinned_by_reference_test": "test_parse_public_input",
        "public_input": json.loads(m.group(1)),
        "expected_u256": want,
    }]
    # 3. worker.rs: the ProofResult of the ignored verify-worker test
    wk = open(os.path.join(REF, "src/settlement/worker.rs")).read()
    mp = re.search(r'public_input:\s*"((?:[^"\\]|\\.)*)"\.to_string\(\),\s*proof:\s*"((?:[^"\\]|\\.)*)"\.to_string\(\)', wk)
    wl = wk.count("\n", 0, mp.start()) + 1
    w_pub, w_proof = json.loads(rust_string(mp.group(1))), json.loads(rust_string(mp.group(2)))
    cm = open(os.path.join(REF, "src/settlement/custom/methods.rs")).read()
    mc = re.search(r'"(\{\\"pi_a\\"(?:[^"\\]|\\.)*)"\.to_string\(\),\s*"((?:[^"\\]|\\.)*)"\.to_string\(\)', cm)
    cl = cm.count("\n", 0, mc.start()) + 1
    c_proof, c_pub = json.loads(rust_string(mc.group(1))), json.loads(rust_string(mc.group(2)))
    assert c_proof == w_proof and c_pub == w_pub, "the two call sites hold the same vector"
    fixtures.append({
        "source": "src/settlement/worker.rs:%d-%d = src/settlement/custom/methods.rs:%d-%d" % (wl, wl + 1, cl, cl + 1),
        "pinned_by_reference_test": None,
        "proof": w_proof,
        "public_input": w_pub,
    })
    # 4. the debug stand-in files (also tests/golden/ref_proof.json / ref_public_input.json)
    fixtures.append({
        "source": "proof/proof.json, proof/public_input.json (src/settlement/worker.rs:56-59)",
        "pinned_by_reference_test": None,
        "proof": json.load(open(os.path.join(REF, "proof/proof.json"))),
        "public_input": json.load(open(os.path.join(REF, "proof/public_input.json"))),
    })
    del lines
    json.dump({"proofs": fixtures, "public_inputs": public_inputs}, open(OUT, "w"), indent=1)
    print("wrote", OUT, "-", len(fixtures), "proofs,", len(public_inputs), "public-input vector")


if __name__ == "__main__":
    main()
