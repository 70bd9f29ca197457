"""Mirror of what eigen-zeth does with a final proof AFTER the prover has answered -- the acceptance grammar the service's output must
satisfy (a10, a11).  Same names, argument meaning and error behaviour as the reference's functions:

* `parse_proof(json_str)`         src/settlement/ethereum/mod.rs:445-474  -> Proof{a: G1Point{x, y}, b: G2Point{x[2], y[2]}, c: G1Point{x, y}};
  every coordinate must be a JSON *string* holding a decimal number that fits 256 bits (`U256::from_dec_str`), anything else is an
  error; the b coordinates are taken IN JSON ORDER, nothing is swapped (:464-467); extra keys (`protocol`, `curve`) are ignored.
* `parse_public_input(json_str)`  src/settlement/ethereum/mod.rs:476-481  -> [U256; 1]: element 0 of a JSON array, a decimal string.
* `ProofResult`                   src/db/mod.rs:63-71: the record `proof_worker` stores under BATCH_PROOF_<n> (src/settlement/worker.rs:179-184).

The reference's own test vectors for these functions are committed under tests/golden/ref_proofs.json and run against this mirror
(tests/test_ref_proofs.py)."""
from __future__ import annotations

import json
from typing import NamedTuple


class ParseError(ValueError):
    """anyhow!("invalid json data") / a serde or U256 parse error in the reference"""


class G1Point(NamedTuple):
    x: int
    y: int


class G2Point(NamedTuple):
    x: tuple
    y: tuple


class Proof(NamedTuple):
    a: G1Point
    b: G2Point
    c: G1Point

    def as_u256_tuple(self):
        """the eight words in the order verifyBatches receives them (zkvm.rs:82-130)"""
        return (self.a.x, self.a.y, self.b.x[0], self.b.x[1], self.b.y[0], self.b.y[1], self.c.x, self.c.y)


def _u256_from_dec_str(v):
    # serde_json `Value::as_str` is None for anything but a string; U256::from_dec_str takes ASCII digits only and fails on overflow
    if not isinstance(v, str):
        raise ParseError("invalid json data")
    if not v or any(ch not in "0123456789" for ch in v):
        raise ParseError("invalid character in decimal string")
    n = int(v)
    if n >> 256:
        raise ParseError("decimal string overflows 256 bits")
    return n


def _index(v, *path):
    # serde_json's Index on a Value never fails: a missing key / wrong type yields Null, which as_str() then rejects
    for k in path:
        if isinstance(k, str):
            v = v.get(k) if isinstance(v, dict) else None
        else:
            v = v[k] if isinstance(v, list) and 0 <= k < len(v) else None
    return v


def parse_proof(json_str: str) -> Proof:
    try:
        v = json.loads(json_str)
    except (ValueError, TypeError) as e:
        raise ParseError(str(e))
    g = lambda *path: _u256_from_dec_str(_index(v, *path))
    return Proof(a=G1Point(g("pi_a", "x"), g("pi_a", "y")),
                 b=G2Point((g("pi_b", "x", 0), g("pi_b", "x", 1)), (g("pi_b", "y", 0), g("pi_b", "y", 1))),
                 c=G1Point(g("pi_c", "x"), g("pi_c", "y")))


def parse_public_input(json_str: str) -> list:
    try:
        v = json.loads(json_str)
    except (ValueError, TypeError) as e:
        raise ParseError(str(e))
    return [_u256_from_dec_str(_index(v, 0))]


def proof_result_json(block_number: int, proof: str, public_input: str, pre_state_root: bytes, post_state_root: bytes) -> str:
    """serde_json::to_string(&ProofResult{..}) (src/db/mod.rs:63-71): the roots serialise as arrays of 32 numbers"""
    if len(pre_state_root) != 32 or len(post_state_root) != 32:
        raise ValueError("state roots are [u8; 32] (src/prover/provider.rs:323-324)")
    return json.dumps({"block_number": int(block_number), "proof": proof, "public_input": public_input,
                       "pre_state_root": list(pre_state_root), "post_state_root": list(post_state_root)}, separators=(",", ":"))
