"""Extract the Groth16 verifying-key constants embedded in the reference's verifier contract artefact.

Run HERE (where /root/reference exists):  python tests/golden/extract_ref_vk.py  -> tests/golden/ref_vk.json

`contracts/EigenZkVM.json` is a Foundry artefact (ABI + bytecode, no Solidity source).  Its deployed bytecode pushes the
verifying key of `verifyTx(Proof, uint256[1])` as literal constants (the on-chain check the settlement layer calls:
src/settlement/ethereum/interfaces/zkvm.rs:82-130).  The recipe: hex-decode
deployedBytecode.object, walk the opcodes skipping PUSH immediates, keep every PUSH24..PUSH32 immediate whose byte offset
falls in the verifying-key window [10400, 11400].  The output is DATA (18 integers + their offsets + the two moduli the
same bytecode pushes); no source text of the reference is copied.  These are the only BN254 values the reference holds
besides the proof fixtures, so they pin oracle/naive_bn254.py and oracle/bn254_pairing.py on reference-held data
(tests/test_ref_vk.py)."""
import json
import os
import sys

REF = "/root/reference/contracts/EigenZkVM.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vk.json")
WINDOW = (10400, 11400)
P_BN254 = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def pushes(code):
    """(offset, n_bytes, value) of every PUSHn in the byte string"""
    i, out = 0, []
    while i < len(code):
        op = code[i]
        if 0x60 <= op <= 0x7F:
            n = op - 0x5F
            out.append((i, n, int.from_bytes(code[i + 1:i + 1 + n], "big")))
            i += 1 + n
        else:
            i += 1
    return out


def main():
    art = json.load(open(REF))
    code = bytes.fromhex(art["deployedBytecode"]["object"][2:])
    allp = pushes(code)
    vk = [(off, n, v) for (off, n, v) in allp if n >= 24 and WINDOW[0] <= off <= WINDOW[1]]
    moduli = {str(off): ("P" if v == P_BN254 else "R") for (off, n, v) in allp if n == 32 and v in (P_BN254, R_BN254)}
    if len(vk) != 18:
        sys.exit("expected 18 constants in the verifying-key window, found %d" % len(vk))
    c = [v for (_o, _n, v) in vk]
    out = {
        "source": "contracts/EigenZkVM.json deployedBytecode.object, PUSH24..PUSH32 immediates at byte offsets %d..%d" % WINDOW,
        "bytecode_bytes": len(code),
        "constants": [{"offset": off, "push_bytes": n, "value": str(v)} for (off, n, v) in vk],
        "moduli_pushes": moduli,
        # push order: one G1 point (x, y); three G2 points, four words each; one G1 point (x, y); one G1 point pushed y-before-x
        "g1": [[str(c[0]), str(c[1])], [str(c[14]), str(c[15])], [str(c[17]), str(c[16])]],
        "g2_words": [[str(v) for v in c[2:6]], [str(v) for v in c[6:10]], [str(v) for v in c[10:14]]],
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
