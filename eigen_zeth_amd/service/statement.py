"""The block statement a chunk proof is bound to.

GenBatchChunksResult reports pre_state_root / post_state_root (proto/prover/v1/prover.proto:80-91; the client stores them
in ProofResult, src/prover/provider.rs:315-330, src/db/mod.rs:63-71).  Round 2's chunk proofs did not depend on them: the
witness was a function of (chain_id, block, chunk) only.  Now the statement

    (chain_id, block number, chunk index, chunk count of the block, pre-state root, post-state root, block hash,
     digest of the block's transaction hashes)

is hashed (SHA-256 over a fixed-width encoding) into four 64-bit limbs, reduced below the Goldilocks modulus, and those
limbs are the STARTING VALUES of trace cells the chunk AIR constrains to public inputs (L_first * (c_i - pub_i),
stark/air.py).  A proof therefore carries the statement limbs as publics[0..k), they are absorbed into its Fiat-Shamir
transcript, and a verifier that knows the block recomputes them: a proof for block n is rejected for block n + 1.

The real zkEVM executor (which would turn the block into an execution trace) is not obtainable offline;
this binds the synthetic witness to the block data the service does have."""
from __future__ import annotations

import hashlib
import struct

P = 0xFFFFFFFF00000001
STATEMENT_TAG = b"zeth-prover/chunk-statement/v1"


def tx_digest(tx_hashes):
    """SHA-256 over the concatenated 32-byte transaction hashes of the block (hex strings, in block order)"""
    h = hashlib.sha256()
    for t in tx_hashes:
        b = bytes.fromhex(t[2:] if t.startswith("0x") else t)
        if len(b) != 32:
            raise ValueError("malformed transaction hash")
        h.update(b)
    return h.digest()


def statement_bytes(chain_id, block, chunk, n_chunks, pre_root, post_root, block_hash=b"", txd=b""):
    pre_root, post_root = bytes(pre_root), bytes(post_root)
    if len(pre_root) != 32 or len(post_root) != 32:
        raise ValueError("state roots must be 32 bytes")
    block_hash = bytes(block_hash).rjust(32, b"\0")
    txd = bytes(txd).rjust(32, b"\0")
    return (STATEMENT_TAG + struct.pack("<QQII", int(chain_id) & 0xFFFFFFFFFFFFFFFF, int(block) & 0xFFFFFFFFFFFFFFFF, int(chunk), int(n_chunks))
            + pre_root + post_root + block_hash + txd)


def statement_limbs(chain_id, block, chunk, n_chunks, pre_root, post_root, block_hash=b"", txd=b"", n=4):
    """n <= 4 field elements (< p) naming the statement"""
    d = hashlib.sha256(statement_bytes(chain_id, block, chunk, n_chunks, pre_root, post_root, block_hash, txd)).digest()
    return [int.from_bytes(d[8 * i:8 * i + 8], "little") % P for i in range(min(n, 4))]
