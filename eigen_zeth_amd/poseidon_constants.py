"""Poseidon parameter tables for the Goldilocks width-12 permutation.

The reference (eigen-zeth) carries no hash constants at all -- its prover is an external, un-pinned
service -- so the tables are *configuration*: `zp_set_constants` (include/
zeth_prover.h) installs any 360 round constants + 12x12 MDS.  This module produces the default.

Round 6: the default round constants are the table of the PUBLIC Goldilocks-Poseidon family (the one the
pil-stark / eigen-zkvm provers are publicly known to share): 360 draws of `gen_range(0..p)` from a
ChaCha8 stream seeded with the u64 0 (`chacha8_round_constants`).  Both anchors SURVEY.md Appendix A/C
recalls from that family are reproduced: the first constant 0xb585f766f2144405 and, with the MDS below,
perm(0^12)[0..4] = 3c18a9786cb0b359 c4055e3364a246c3 7953db0ab48808f4 c71603f33a1144ca
(tests/test_poseidon_constants.py on the CPU, tests/test_gpu_parity.py through the C-ABI).  The MDS is the
circulant-plus-diagonal matrix: circulant first row [17,15,41,16,2,28,13,13,39,18,34,20] plus diag
[8,0,...,0], i.e. effective first row [25,15,41,...].  The table of rounds 1-5 (Poseidon reference
Grain-LFSR with (field=1, sbox=0, n=64, t=12, R_F=8, R_P=22)) stays available as
`grain_goldilocks_round_constants()`.

The generator itself is anchored to a published value: with (1,0,254,3,8,57) over the BN254 scalar
field its first output is 0x0ee9a592...cd8e6e (tests/test_poseidon_constants.py).
"""
from __future__ import annotations

GL_P = 0xFFFFFFFF00000001
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]


class _Grain:
    def __init__(self, field, sbox, n, t, rf, rp):
        bits = []
        for val, width in ((field, 2), (sbox, 4), (n, 12), (t, 12), (rf, 10), (rp, 10)):
            bits += [(val >> (width - 1 - i)) & 1 for i in range(width)]
        bits += [1] * 30
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        # self-shrinking: a 1 selects the next bit, a 0 drops it
        nb = self._step()
        while nb == 0:
            self._step()
            nb = self._step()
        return self._step()

    def integer(self, nbits):
        v = 0
        for _ in range(nbits):
            v = (v << 1) | self.bit()
        return v


def grain_round_constants(field, sbox, n, t, rf, rp, prime, count=None):
    g = _Grain(field, sbox, n, t, rf, rp)
    count = (rf + rp) * t if count is None else count
    out = []
    while len(out) < count:
        v = g.integer(n)
        if v < prime:
            out.append(v)
    return out


# ---- Poseidon over the BN254 scalar field (BN128-hash mode of the final STARK): x^5, 8 full rounds,
# partial rounds per width as in the published parameter sets (t = 2 .. 17)
BN254_RP = dict(zip(range(2, 18), [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]))


def bn254_poseidon_params(t):
    """(round constants [(8 + R_P) * t], MDS rows [t][t], R_P) from the Poseidon reference procedure: Grain LFSR
    (1, 0, 254, t, 8, R_P); round constants by rejection sampling, then 2t further field elements x_i, y_j (reduced mod r,
    distinct) and the Cauchy matrix M[i][j] = 1 / (x_i + y_j).  For t = 3 this is the widely published instance: first
    constant 0x0ee9a592..., M[0][0] = 0x109b7f41..., poseidon([1, 2]) = 0x115cc0f5...189a (tests/test_poseidon_constants.py)."""
    rp = BN254_RP[t]
    g = _Grain(1, 0, 254, t, 8, rp)
    rc = []
    while len(rc) < (8 + rp) * t:
        v = g.integer(254)
        if v < BN254_R:
            rc.append(v)
    while True:
        xy = [g.integer(254) % BN254_R for _ in range(2 * t)]
        if len(set(xy)) == 2 * t and all((xy[i] + xy[t + j]) % BN254_R for i in range(t) for j in range(t)):
            break
    mds = [[pow((xy[i] + xy[t + j]) % BN254_R, -1, BN254_R) for j in range(t)] for i in range(t)]
    return rc, mds, rp


def grain_goldilocks_round_constants():
    """the default table of rounds 1-5: Grain LFSR (1, 0, 64, 12, 8, 22)"""
    return grain_round_constants(1, 0, 64, 12, 8, 22, GL_P)


# ---- the public family's table: a ChaCha8 stream, seeded the way a 64-bit seed is widened to a 256-bit key by a PCG32
# sequence, drawn as uniform integers below p by the widening-multiply rejection rule
_M32 = 0xFFFFFFFF
_M64 = 0xFFFFFFFFFFFFFFFF


def _pcg32_key(seed64):
    """eight little-endian u32 words of a PCG32 (XSH-RR) sequence started from `seed64`: the 256-bit key"""
    st, key = seed64 & _M64, []
    for _ in range(8):
        st = (st * 6364136223846793005 + 11634580027462260723) & _M64
        xs, rot = (((st >> 18) ^ st) >> 27) & _M32, st >> 59
        key.append(((xs >> rot) | (xs << (32 - rot))) & _M32 if rot else xs)
    return key


def _chacha_block(key, counter, rounds):
    """one 16-word ChaCha block: 64-bit block counter in words 12-13, stream id 0 in words 14-15"""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + key + [counter & _M32, counter >> 32, 0, 0]
    s = list(init)

    def quarter(a, b, c, d):
        for x, y, z, n in ((a, b, d, 16), (c, d, b, 12), (a, b, d, 8), (c, d, b, 7)):
            s[x] = (s[x] + s[y]) & _M32
            v = s[z] ^ s[x]
            s[z] = ((v << n) | (v >> (32 - n))) & _M32

    for _ in range(rounds // 2):
        for a in range(4):
            quarter(a, 4 + a, 8 + a, 12 + a)
        for a in range(4):
            quarter(a, 4 + (a + 1) % 4, 8 + (a + 2) % 4, 12 + (a + 3) % 4)
    return [(s[i] + init[i]) & _M32 for i in range(16)]


def chacha8_round_constants(seed64=0, count=360, prime=GL_P):
    """`count` uniform draws below `prime` (a 64-bit prime with its top bit set) from ChaCha8 keyed by `seed64`: 64-bit words are
    (low, high) pairs of consecutive output words; a word v is accepted when the low half of v * prime is <= prime - 1 and the
    draw is the high half."""
    assert prime >> 63 == 1
    key, words, ctr, out = _pcg32_key(seed64), [], 0, []
    while len(out) < count:
        if len(words) < 2:
            words += _chacha_block(key, ctr, 8)
            ctr += 1
        v = words[0] | (words[1] << 32)
        del words[:2]
        m = v * prime
        if (m & _M64) <= prime - 1:
            out.append(m >> 64)
    return out


def default_round_constants():
    """360 Goldilocks constants, round-major: rc[r*12 + i] -- the public family's table (module docstring)"""
    return chacha8_round_constants(0)


def default_mds():
    """flat row-major 12x12: out[r] = sum_j mds[r*12+j] * in[j];  coefficient of in[j] in out[r] is
    circ[(j-r) mod 12] (+ diag[r] on the diagonal)."""
    m = [0] * 144
    for r in range(12):
        for j in range(12):
            m[r * 12 + j] = MDS_CIRC[(j - r) % 12] + (MDS_DIAG[r] if j == r else 0)
    return m


def write_inc(path):
    rc = default_round_constants()
    mds = default_mds()
    with open(path, "w") as f:
        f.write("// generated by eigen_zeth_amd/poseidon_constants.py (ChaCha8 stream seeded with 0: the public Goldilocks-Poseidon table) -- do not edit\n")
        f.write("static const uint64_t ZP_POSEIDON_DEFAULT_RC[360] = {\n")
        for i in range(0, 360, 4):
            f.write("    " + ", ".join("0x%016xULL" % v for v in rc[i:i + 4]) + ",\n")
        f.write("};\nstatic const uint64_t ZP_POSEIDON_DEFAULT_MDS[144] = {\n")
        for r in range(12):
            f.write("    " + ", ".join("%d" % v for v in mds[r * 12:(r + 1) * 12]) + ",\n")
        f.write("};\n")


if __name__ == "__main__":
    import os
    write_inc(os.path.join(os.path.dirname(__file__), "csrc", "poseidon_default_table.inc"))
