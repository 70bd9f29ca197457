"""External anchors of the constant generators.

The reference holds no hash constants.  Two chains in this repo's hash stack are pinned to values published outside it:
(1) BN254 side: the Poseidon reference Grain-LFSR stream with the parameters of the widely published BN254 t = 3 instance
(field 1, s-box 0, n = 254, t = 3, R_F = 8, R_P = 57) yields that instance's first round constants and its hash of [1, 2];
(2) Goldilocks side (round 6): a ChaCha8 stream seeded with 0 yields the public Goldilocks-Poseidon family's first round
constant, and the permutation over that table and the circulant + diagonal linear layer maps 0^12 to the family's recalled
output words -- two independent generators (product: poseidon_constants.chacha8_round_constants; checker:
oracle/chacha8_table.py) and two independent permutations (big-int definition, C restatement) agree on both anchors.  That
table is the one the C-ABI library compiles in (csrc/poseidon_default_table.inc)."""
import os
import re

from eigen_zeth_amd import poseidon_constants as PC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# first four round constants of the published BN254 x^5 t=3 instance (8 full + 57 partial rounds)
BN254_T3_FIRST = [
    0x0EE9A592BA9A9518D05986D656F40C2114C4993C11BB29938D21D47304CD8E6E,
    0x00F1445235F2148C5986587169FC1BCD887B08D4D00868DF5696FFF40956E864,
    0x08DFF3487E8AC99E1F29A058D0FA80B930C728730B7AB36CE879F3890ECF73F5,
    0x2F27BE690FDAEE46C3CE28F7532B13C856C35342C84BDA6E20966310FADC01D0,
]


def test_grain_lfsr_reproduces_published_bn254_t3_constants():
    got = PC.grain_round_constants(1, 0, 254, 3, 8, 57, PC.BN254_R, count=4)
    assert got == BN254_T3_FIRST


def test_grain_stream_is_prefix_stable_and_in_range():
    a = PC.grain_round_constants(1, 0, 254, 3, 8, 57, PC.BN254_R, count=10)
    b = PC.grain_round_constants(1, 0, 254, 3, 8, 57, PC.BN254_R)
    assert len(b) == (8 + 57) * 3 and b[:10] == a
    assert all(0 <= v < PC.BN254_R for v in b)


def test_grain_goldilocks_table_of_rounds_1_to_5_is_still_reachable():
    rc = PC.grain_goldilocks_round_constants()
    assert len(rc) == 360 and all(0 <= v < PC.GL_P for v in rc)
    # the (1,0,64,12,8,22) stream starts with these two values
    assert rc[0] == 0x13DCF33ABA214F46 and rc[1] == 0x30B3B654A1DA6D83


def test_chacha8_seed0_table_hits_the_public_familys_anchors():
    """SURVEY.md Appendix A / C record two values of the public Goldilocks-Poseidon family: the first round constant and
    perm(0^12)[0..4].  The checker's own generator (oracle/chacha8_table.py) reproduces the first, its big-int permutation and
    its C permutation reproduce the second, and the product's generator yields the very same 360 words."""
    import numpy as np
    from oracle import chacha8_table as CT
    from oracle import naive as NV
    from oracle import oracle as O
    rc = CT.round_constants(0)
    assert rc[0] == CT.ANCHOR_FIRST_CONSTANT == 0xB585F766F2144405
    assert len(rc) == 360 and all(0 <= v < PC.GL_P for v in rc) and len(set(rc)) == 360
    assert rc == PC.chacha8_round_constants(0) == PC.default_round_constants()
    mds = PC.default_mds()
    out = NV.poseidon_perm([0] * 12, rc, mds)
    assert out[:4] == CT.ANCHOR_PERM_ZERO == [0x3C18A9786CB0B359, 0xC4055E3364A246C3, 0x7953DB0AB48808F4, 0xC71603F33A1144CA]
    got = O.poseidon_perm(np.zeros((1, 12), dtype=np.uint64), np.array(rc, dtype=np.uint64), np.array(mds, dtype=np.uint64))
    assert [int(v) for v in got[0]] == out
    # two more vectors of the public family's test suite (recalled in round 6 before they were computed): non-zero inputs exercise every column of the
    # linear layer and the S-box of all twelve lanes, which perm(0) alone does not
    assert NV.poseidon_perm(list(range(12)), rc, mds)[:4] == CT.ANCHOR_PERM_COUNTING
    assert NV.poseidon_perm([PC.GL_P - 1] * 12, rc, mds)[0] == CT.ANCHOR_PERM_MINUS_ONE_WORD0
    sts = np.array([list(range(12)), [PC.GL_P - 1] * 12], dtype=np.uint64)
    got = O.poseidon_perm(sts, np.array(rc, dtype=np.uint64), np.array(mds, dtype=np.uint64))
    assert [int(v) for v in got[0, :4]] == CT.ANCHOR_PERM_COUNTING and int(got[1, 0]) == CT.ANCHOR_PERM_MINUS_ONE_WORD0
    # neither anchor survives another seed, round count or the Grain table: the match is not an accident of the linear layer
    assert CT.round_constants(1)[0] != CT.ANCHOR_FIRST_CONSTANT
    assert CT.uniform_below(CT.ChaCha(CT.key_from_u64(0), 12), CT.P) != CT.ANCHOR_FIRST_CONSTANT
    assert NV.poseidon_perm([0] * 12, PC.grain_goldilocks_round_constants(), mds)[:4] != CT.ANCHOR_PERM_ZERO


def test_default_goldilocks_table():
    rc = PC.default_round_constants()
    assert len(rc) == 360 and all(0 <= v < PC.GL_P for v in rc)
    assert rc[0] == 0xB585F766F2144405 and rc[1] == 0x7746A55F43921AD7
    mds = PC.default_mds()
    # effective first row = circulant first row + diagonal term: [17+8, 15, 41, ...]
    assert mds[:12] == [25, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    for r in range(12):
        for j in range(12):
            assert mds[r * 12 + j] == PC.MDS_CIRC[(j - r) % 12] + (8 if r == j == 0 else 0)


def test_compiled_in_table_is_the_generated_one():
    path = os.path.join(ROOT, "eigen_zeth_amd", "csrc", "poseidon_default_table.inc")
    text = open(path).read()
    rc_txt, mds_txt = text.split("ZP_POSEIDON_DEFAULT_MDS")
    rc = [int(h, 16) for h in re.findall(r"0x([0-9a-f]{16})ULL", rc_txt)]
    assert rc == PC.default_round_constants()
    mds = [int(v) for v in re.findall(r"\b(\d+)\b", mds_txt.split("{", 1)[1])]
    assert mds == PC.default_mds()


def test_bn254_t3_instance_reproduces_the_published_hash_vector():
    """the whole chain -- Grain round constants, the Cauchy MDS drawn from the continued stream, the permutation -- against
    the published value poseidon([1, 2]) of the BN254 x^5 t = 3 instance (the only externally pinned hash in the repo)"""
    from oracle import naive as NV
    rc, mds, rp = PC.bn254_poseidon_params(3)
    assert rp == 57 and rc[:4] == BN254_T3_FIRST and len(rc) == 65 * 3
    assert mds[0][0] == 0x109B7F411BA0E4C9B2B70CAF5C36A7B194BE7C11AD24378BFEDB68592BA8118B
    assert NV.poseidon_bn254_hash([1, 2], rc, mds, rp) == 0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A
    # the checker's C restatement (4 x 64-bit Montgomery words) hits the same published value, and agrees with the
    # big-int definition on random states of both widths
    import random
    from oracle import oracle as O
    O.p254_set(3, rp, rc, mds)
    assert O.p254_perm([[0, 1, 2]], 3)[0][0] == 0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A
    rnd = random.Random(17)
    for t in (3, 17):
        rc_t, mds_t, rp_t = PC.bn254_poseidon_params(t)
        O.p254_set(t, rp_t, rc_t, mds_t)
        sts = [[rnd.randrange(PC.BN254_R) for _ in range(t)] for _ in range(2)] + [[0] * t, [PC.BN254_R - 1] * t]
        assert O.p254_perm(sts, t) == [NV.poseidon_bn254_perm(st, rc_t, mds_t, rp_t) for st in sts]


def test_bn254_t17_and_t5_instances_reproduce_published_vectors():
    """round 6: two more vectors of the public hash family, written down from memory and THEN computed -- poseidon([1, 2, 3, 4]) of the t = 5
    instance (R_P = 60) and poseidon([1, ..., 16]) of the t = 17 instance (R_P = 68), the width this repo's BN128 mode uses for every tree, sponge
    and in-circuit gadget.  They pin, for that width, the Grain stream, the rejection sampling, the Cauchy matrix drawn from the continued
    stream, the round numbers and the schedule -- in the big-integer definition and in the checker's C restatement"""
    from oracle import naive as NV
    from oracle import oracle as O
    rc, mds, rp = PC.bn254_poseidon_params(5)
    assert rp == 60
    assert NV.poseidon_bn254_perm([0, 1, 2, 3, 4], rc, mds, rp)[0] == 0x299C867DB6C1FDD79DCEFA40E4510B9837E60EBB1CE0663DBAA525DF65250465
    rc, mds, rp = PC.bn254_poseidon_params(17)
    want = 9989051620750914585850546081941653841776809718687451684622678807385399211877
    assert rp == 68
    assert NV.poseidon_bn254_perm([0] + list(range(1, 17)), rc, mds, rp)[0] == want
    O.p254_set(17, rp, rc, mds)
    assert O.p254_perm([[0] + list(range(1, 17))], 17)[0][0] == want


def test_bn254_t17_parameters_are_well_formed():
    rc, mds, rp = PC.bn254_poseidon_params(17)
    assert rp == 68 and len(rc) == 76 * 17 and len(mds) == 17 and all(len(r) == 17 for r in mds)
    assert all(0 < v < PC.BN254_R for row in mds for v in row) and len({v for row in mds for v in row}) == 289


def test_generated_mds_rows_are_the_committed_ones(tmp_path):
    """csrc/poseidon_mds_asm.inc is generated (tools/gen_mds_asm.py) from the same circulant the tables use"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "eigen_zeth_amd", "csrc", "poseidon_mds_asm.inc")
    out = str(tmp_path / "poseidon_mds_asm.inc")          # never into the tree: the include is a prerequisite of every object of the library
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_mds_asm.py"), out], stdout=subprocess.DEVNULL)
    assert open(out).read() == open(inc).read()
    src = open(os.path.join(root, "tools", "gen_mds_asm.py")).read()
    circ = [int(v) for v in re.search(r"CIRC = \[([^\]]*)\]", src).group(1).split(",")]
    m = PC.default_mds()
    assert [m[j] for j in range(12)] == [circ[j] + (8 if j == 0 else 0) for j in range(12)]
    assert all(m[i * 12 + j] == circ[(j - i) % 12] + (8 if i == j == 0 else 0) for i in range(12) for j in range(12))


def test_partial_rounds_three_at_a_time_equal_the_textbook_schedule():
    """the algebra csrc/poseidon.hip's partial3_default relies on (round 6), in big integers: with t = the state after the first S-box of a
    block, Z = clear element 0, c1..c3 the constants after rounds r..r+2 --  x1 = (M t)_0 + c1_0;  x2 = (M Z M t)_0 + M_00 y1 + (M Z c1 + c2)_0;
    next state = (M Z)^2 M t + (M Z M e0) y1 + (M e0) y2 + (M Z)^2 c1 + M Z c2 + c3 -- equals three textbook rounds; and every integer
    matrix entry stays below 2^21, so that the kernel's 64-bit sums over 32-bit halves (14 terms per row) cannot overflow"""
    import random
    p = PC.GL_P
    rc = [int(v) for v in PC.default_round_constants()]
    m = [int(v) for v in PC.default_mds()]
    M = [m[i * 12:(i + 1) * 12] for i in range(12)]
    MZ = [[0 if j == 0 else M[i][j] for j in range(12)] for i in range(12)]
    mm = lambda A, B: [[sum(A[i][k] * B[k][j] for k in range(12)) for j in range(12)] for i in range(12)]
    mv = lambda A, v: [sum(A[i][j] * v[j] for j in range(12)) % p for i in range(12)]
    A2 = mm(MZ, M)
    A3 = mm(MZ, A2)
    assert max(max(r) for r in A3) < 1 << 21 and max(max(r) for r in A2) < 1 << 14
    assert all(sum(A3[i]) + A2[i][0] + M[i][0] < 1 << 25 for i in range(12))        # row sums: (2^25 * 2^32) per half sum, far below 2^64
    rnd = random.Random(6)
    for r in range(4, 25, 3):
        c1, c2, c3 = (rc[(r + k) * 12:(r + k + 1) * 12] for k in (1, 2, 3))
        s = [rnd.randrange(p) for _ in range(12)]
        want = list(s)
        for k in range(3):                            # textbook: S-box on element 0, matrix, constants of the next round
            want[0] = pow(want[0], 7, p)
            want = [(a + b) % p for a, b in zip(mv(M, want), rc[(r + k + 1) * 12:(r + k + 2) * 12])]
        t = [pow(s[0], 7, p)] + s[1:]
        y1 = pow((mv(M, t)[0] + c1[0]) % p, 7, p)
        y2 = pow((mv(A2, t)[0] + M[0][0] * y1 + mv(MZ, c1)[0] + c2[0]) % p, 7, p)
        k3 = [(a + b + c) % p for a, b, c in zip(mv(MZ, mv(MZ, c1)), mv(MZ, c2), c3)]
        got = [(mv(A3, t)[i] + A2[i][0] * y1 + M[i][0] * y2 + k3[i]) % p for i in range(12)]
        assert got == want


def test_default_root_of_unity_reproduces_the_public_cpp_librarys_table_of_two_adic_roots():
    """round 6: the public C++ Goldilocks library of the pil-stark prover family (the lineage of the external prover eigen-zeth calls) ships a
    table W[0..32] of 2^k-th roots of unity.  Its 33 words, written down from memory and THEN compared: every one equals
    w32^(2^(32 - k)) for this repo's DEFAULT 2^32-th root 1753635133440165772 = 7^((p - 1) / 2^32) -- which is also the public Rust
    Goldilocks field's two-adic generator.  SURVEY.md section 8a lists the root as an unpinned choice between two candidates; the other
    candidate (7277203076849721926) does not produce this table.  The coset shift 49 = 7^2 is the same family's."""
    from eigen_zeth_amd import native
    p = PC.GL_P
    W = [1, 18446744069414584320, 281474976710656, 18446744069397807105, 17293822564807737345, 70368744161280, 549755813888,
         17870292113338400769, 13797081185216407910, 1803076106186727246, 11353340290879379826, 455906449640507599,
         17492915097719143606, 1532612707718625687, 16207902636198568418, 17776499369601055404, 6115771955107415310,
         12380578893860276750, 9306717745644682924, 18146160046829613826, 3511170319078647661, 17654865857378133588,
         5416168637041100469, 16905767614792059275, 9713644485405565297, 5456943929260765144, 17096174751763063430,
         1213594585890690845, 6414415596519834757, 16116352524544190054, 9123114210336311365, 4614640910117430873,
         1753635133440165772]
    w32 = native.ROOT32_DEFAULT
    assert w32 == W[32] == pow(7, (p - 1) >> 32, p)
    assert all(pow(w32, 1 << (32 - k), p) == W[k] for k in range(33))
    assert W[2] == 1 << 48 and W[1] == p - 1
    alt = native.ROOT32_ALT
    assert pow(alt, 1 << 32, p) == 1 and pow(alt, 1 << 31, p) == p - 1 and any(pow(alt, 1 << (32 - k), p) != W[k] for k in range(3, 33))


def test_public_familys_three_test_vectors_in_full():
    """all twelve output words of the public Goldilocks-Poseidon family's three test vectors (zeros, 0..11, p - 1 everywhere) -- 36 words from
    memory -- against the big-integer definition and the checker's C restatement over the generated default table"""
    import numpy as np
    from oracle import chacha8_table as CT
    from oracle import naive as NV
    from oracle import oracle as O
    rc, mds = [int(v) for v in PC.default_round_constants()], [int(v) for v in PC.default_mds()]
    states = {"zero": [0] * 12, "counting": list(range(12)), "minus_one": [PC.GL_P - 1] * 12}
    for name, st in states.items():
        assert NV.poseidon_perm(st, rc, mds) == CT.ANCHOR_FULL[name], name
        got = O.poseidon_perm(np.array([st], dtype=np.uint64), np.array(rc, dtype=np.uint64), np.array(mds, dtype=np.uint64))[0]
        assert [int(v) for v in got] == CT.ANCHOR_FULL[name], name
