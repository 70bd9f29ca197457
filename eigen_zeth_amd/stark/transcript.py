"""Fiat-Shamir transcript: Poseidon-12 sponge (rate 8, capacity 4) over Goldilocks.
absorb() queues elements; squeeze() first absorbs everything queued in blocks of 8 (overwriting the
rate, zero padded; at least one permutation), then hands out rate elements, permuting again when
the 8 are used up.  The permutation is supplied by the backend (GPU: zp_poseidon_perm)."""
P = 0xFFFFFFFF00000001


class Transcript:
    def __init__(self, perm):
        self.perm = perm          # callable: list[12] -> list[12]
        self.state = [0] * 12
        self.pending = []
        self.out = []

    def absorb(self, vals):
        self.pending += [int(v) % P for v in vals]
        self.out = []

    def _flush(self):
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
                self._flush()
            res.append(self.out.pop(0))
        return res

    def challenge_e3(self):
        return self.squeeze(3)

    def indices(self, count, bits):
        vals = self.squeeze(count)
        return [v & ((1 << bits) - 1) for v in vals]
