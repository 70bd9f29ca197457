"""TEST INFRASTRUCTURE (checker side; the product never imports this).

An independent restatement of how the PUBLIC Goldilocks-Poseidon round-constant table is made, so that the table the
product compiles in (`eigen_zeth_amd/poseidon_constants.py::chacha8_round_constants`, a different piece of code) is pinned
from outside:

    ChaCha8 (8 rounds = 4 double rounds; 64-bit block counter from 0; stream id 0)
      keyed by the 32 bytes a PCG32 XSH-RR sequence yields when started from the 64-bit seed 0
      (multiplier 6364136223846793005, increment 11634580027462260723, state advanced BEFORE each output);
    360 draws "uniform in 0..p": take a 64-bit word v (low 32-bit word first), form the 128-bit product v * p, accept when
      its low half <= p - 1 and return its high half.

Nothing of this is in /root/reference (the reference holds no hash constants).  The two anchors SURVEY.md
Appendix A / C record from the public family are what pins it: first constant 0xb585f766f2144405, and -- with the
circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag [8,0..] linear layer -- perm(0^12)[0..4] =
3c18a9786cb0b359 c4055e3364a246c3 7953db0ab48808f4 c71603f33a1144ca.  Both reproduce (tests/test_poseidon_constants.py).
"""
import struct

P = 0xFFFFFFFF00000001

ANCHOR_FIRST_CONSTANT = 0xB585F766F2144405
ANCHOR_PERM_ZERO = [0x3C18A9786CB0B359, 0xC4055E3364A246C3, 0x7953DB0AB48808F4, 0xC71603F33A1144CA]
# Two more vectors of the public family's own test suite, written down from memory in round 6 BEFORE the permutation was run on them (they are not in
# SURVEY.md; same standing as the anchors above: recalled, then reproduced): perm(0, 1, ..., 11)[0..4] and perm(p - 1, ..., p - 1)[0].
ANCHOR_PERM_COUNTING = [0xD64E1E3EFC5B8E9E, 0x53666633020AAA47, 0xD40285597C6A8825, 0x613A4F81E81231D2]
ANCHOR_PERM_MINUS_ONE_WORD0 = 0xBE0085CFC57A8357
# Round 6, second half: the public family's three test vectors IN FULL (12 words each: inputs 0^12, 0..11, (p-1)^12), again written down
# from memory and then compared (tests/test_poseidon_constants.py, tests/test_gpu_parity.py): 36 of 36 words
ANCHOR_FULL = {
    "zero": [0x3C18A9786CB0B359, 0xC4055E3364A246C3, 0x7953DB0AB48808F4, 0xC71603F33A1144CA, 0xD7709673896996DC, 0x46A84E87642F44ED,
             0xD032648251EE0B3C, 0x1C687363B207DF62, 0xDF8565563E8045FE, 0x40F5B37FF4254DAE, 0xD070F637B431067C, 0x1792B1C4342109D7],
    "counting": [0xD64E1E3EFC5B8E9E, 0x53666633020AAA47, 0xD40285597C6A8825, 0x613A4F81E81231D2, 0x414754BFEBD051F0, 0xCB1F8980294A023F,
                 0x6EB2A9E4D54A9D0F, 0x1902BC3AF467E056, 0xF045D5EAFDC6021F, 0xE4150F77CAAA3BE5, 0xC9BFD01D39B50CCE, 0x5C0A27FCB0E1459B],
    "minus_one": [0xBE0085CFC57A8357, 0xD95AF71847D05C09, 0xCF55A13D33C1C953, 0x95803A74F4530E82, 0xFCD99EB30A135DF1, 0xE095905E913A3029,
                  0xDE0392461B42919B, 0x7D3260E24E81D031, 0x10D3D0465D9DEAA0, 0xA87571083DFC2A47, 0xE18263681E9958F8, 0xE28E96F1AE5E60D3],
}


def _rotl32(x, n):
    return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF


class ChaCha:
    """keystream as 32-bit words, block after block"""

    def __init__(self, key32, rounds):
        self.key = struct.unpack("<8I", key32)
        self.rounds = rounds
        self.counter = 0
        self.pending = []

    def _block(self):
        c = self.counter
        self.counter += 1
        x0 = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, *self.key, c & 0xFFFFFFFF, (c >> 32) & 0xFFFFFFFF, 0, 0]
        x = x0[:]
        for _ in range(self.rounds >> 1):
            for a, b, c_, d in ((0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15),
                                (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14)):
                x[a] = (x[a] + x[b]) & 0xFFFFFFFF
                x[d] = _rotl32(x[d] ^ x[a], 16)
                x[c_] = (x[c_] + x[d]) & 0xFFFFFFFF
                x[b] = _rotl32(x[b] ^ x[c_], 12)
                x[a] = (x[a] + x[b]) & 0xFFFFFFFF
                x[d] = _rotl32(x[d] ^ x[a], 8)
                x[c_] = (x[c_] + x[d]) & 0xFFFFFFFF
                x[b] = _rotl32(x[b] ^ x[c_], 7)
        return [(u + v) & 0xFFFFFFFF for u, v in zip(x, x0)]

    def word32(self):
        if not self.pending:
            self.pending = self._block()
        return self.pending.pop(0)

    def word64(self):
        lo = self.word32()
        return lo | (self.word32() << 32)


def key_from_u64(seed):
    state, out = seed, b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) % (1 << 64)
        xorshifted = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
        rot = state >> 59
        out += struct.pack("<I", ((xorshifted >> rot) | (xorshifted << ((32 - rot) % 32))) & 0xFFFFFFFF)
    return out


def uniform_below(rng, n):
    zone = ((n << (64 - n.bit_length())) - 1) % (1 << 64)
    while True:
        hi, lo = divmod(rng.word64() * n, 1 << 64)
        if lo <= zone:
            return hi


def round_constants(seed=0, count=360):
    rng = ChaCha(key_from_u64(seed), 8)
    return [uniform_below(rng, P) for _ in range(count)]
