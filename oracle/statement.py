"""oracle/statement.py -- the verifier's side of the block binding (TEST INFRASTRUCTURE; imports nothing from the product).

A chunk proof's first public inputs must be the limbs of the statement of ITS block: SHA-256 over
tag | chain_id u64 | block u64 | chunk u32 | n_chunks u32 | pre_state_root[32] | post_state_root[32] | block_hash[32] |
tx_digest[32] (little-endian integers), cut into 64-bit little-endian words reduced mod p.  The roots are what
GenBatchChunksResult carries (reference: proto/prover/v1/prover.proto:80-91, consumed at src/prover/provider.rs:315-330).
The verifier takes the block data from ITS source (the L2 node), never from the proof."""
import hashlib

P = 0xFFFFFFFF00000001
TAG = b"zeth-prover/chunk-statement/v1"


def tx_digest(tx_hashes):
    return hashlib.sha256(b"".join(bytes.fromhex(t[2:] if t.startswith("0x") else t) for t in tx_hashes)).digest()


def limbs(chain_id, block, chunk, n_chunks, pre_root, post_root, block_hash=b"", txd=b"", n=4):
    msg = TAG
    msg += int(chain_id).to_bytes(8, "little") + int(block).to_bytes(8, "little")
    msg += int(chunk).to_bytes(4, "little") + int(n_chunks).to_bytes(4, "little")
    msg += bytes(pre_root) + bytes(post_root) + bytes(block_hash).rjust(32, b"\0") + bytes(txd).rjust(32, b"\0")
    d = hashlib.sha256(msg).digest()
    return [int.from_bytes(d[8 * i:8 * i + 8], "little") % P for i in range(n)]


def bound_to(proof, chain_id, block, chunk, n_chunks, pre_root, post_root, block_hash=b"", txd=b"", n=4):
    """True iff the proof's leading public inputs are the limbs of this statement (the STARK itself is checked by
    stark_verify.verify, which absorbs the publics into the transcript)"""
    want = limbs(chain_id, block, chunk, n_chunks, pre_root, post_root, block_hash, txd, n)
    return [int(v) for v in proof["publics"][:n]] == want
