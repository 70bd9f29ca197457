"""Fiat-Shamir transcript: Poseidon-12 sponge (rate 8, capacity 4) over Goldilocks.
absorb() queues elements; squeeze() first absorbs everything queued in blocks of 8 (overwriting the
rate, zero padded; at least one permutation), then hands out rate elements, permuting again when
the 8 are used up.  The permutation is supplied by the backend (GPU: zp_poseidon_perm)."""
P = 0xFFFFFFFF00000001


class Transcript:
    def __init__(self, perm, sponge=None):
        """perm: list[12] -> list[12].  sponge (optional, GPU: zp_poseidon_sponge): (state, blocks, extra) -> (state, rates) runs a
        whole absorb / squeeze step in one call; the result is the same as permutation by permutation"""
        self.perm = perm
        self.sponge = sponge
        self.state = [0] * 12
        self.pending = []
        self.out = []

    def absorb(self, vals):
        self.pending += [int(v) % P for v in vals]
        self.out = []

    def absorb_root(self, root):
        """a Merkle root of the Goldilocks-hash mode: four field elements"""
        self.absorb(root)

    def _flush(self, want=8):
        """absorb what is queued (one permutation if nothing is), then make at least `want` output elements available"""
        if self.sponge is not None:
            blocks = [self.pending[i:i + 8] + [0] * (8 - len(self.pending[i:i + 8])) for i in range(0, len(self.pending), 8)]
            self.pending = []
            self.state, rates = self.sponge(self.state, blocks, max(0, (want + 7) // 8 - 1))
            self.out = [v for r in rates for v in r]
            return
        if not self.pending:
            self.state = self.perm(self.state)
        while self.pending:
            blk, self.pending = self.pending[:8], self.pending[8:]
            blk += [0] * (8 - len(blk))
            self.state = self.perm(blk + self.state[8:])
        self.out = list(self.state[:8])

    def squeeze(self, n):
        res = []
        while len(res) < n:
            if self.pending or not self.out:
                self._flush(n - len(res))
            res.append(self.out.pop(0))
        return res

    def challenge_e3(self):
        return self.squeeze(3)

    def indices(self, count, bits):
        vals = self.squeeze(count)
        return [v & ((1 << bits) - 1) for v in vals]


R_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_M64 = (1 << 64) - 1


class TranscriptBN128:
    """Fiat-Shamir transcript of the BN128-hash mode (the last STARK before the Groth16 wrap, whose verifier lives in a
    circuit over the BN254 scalar field): a Poseidon-BN254 sponge of width 17 -- element 0 is the capacity, 1..16 the rate.
    absorb(vals) packs Goldilocks values three to a field element (a + b 2^64 + c 2^128; every call is padded on its own);
    absorb_root takes a Merkle root, which IS one field element.  squeeze() first absorbs what is queued in blocks of 16
    that overwrite the rate (zero padded; one permutation even when nothing is queued), then hands out Goldilocks values:
    the three low 64-bit words of rate elements 1..16, each reduced mod p (48 values per permutation).
    The permutation is supplied by the backend (GPU: zp_poseidon_bn254_perm, t = 17).  Own convention: parity with the
    external prover is unpinned."""

    def __init__(self, perm17):
        self.perm = perm17        # callable: list[17] -> list[17]
        self.state = [0] * 17
        self.pending = []
        self.out = []

    def absorb(self, vals):
        v = [int(x) % P for x in vals]
        v += [0] * (-len(v) % 3)
        self.pending += [v[i] + (v[i + 1] << 64) + (v[i + 2] << 128) for i in range(0, len(v), 3)]
        self.out = []

    def absorb_root(self, root):
        assert len(root) == 1 and 0 <= int(root[0]) < R_BN254
        self.pending.append(int(root[0]))
        self.out = []

    def _flush(self):
        if not self.pending:
            self.state = self.perm(self.state)
        while self.pending:
            blk, self.pending = self.pending[:16], self.pending[16:]
            blk += [0] * (16 - len(blk))
            self.state = self.perm([self.state[0]] + blk)
        self.out = [((e >> (64 * k)) & _M64) % P for e in self.state[1:] for k in range(3)]

    def squeeze(self, n):
        res = []
        while len(res) < n:
            if self.pending or not self.out:
                self._flush()
            res.append(self.out.pop(0))
        return res

    def challenge_e3(self):
        return self.squeeze(3)

    def indices(self, count, bits):
        return [v & ((1 << bits) - 1) for v in self.squeeze(count)]
