"""Minimal BN254 (alt_bn128) group arithmetic on Python ints -- only what the final-proof encoder
needs to emit well-formed curve points in the grammar src/settlement/ethereum/mod.rs:445-481
(parse_proof / parse_public_input) accepts.  Not a pairing library, not the MSM kernel."""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
G1 = (1, 2)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _inv(a):
    return pow(a, P - 2, P)


# ---- F_p2 = F_p[u]/(u^2+1), elements (c0, c1)
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_inv(a):
    d = _inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * d % P, (-a[1]) * d % P)


class _Ops1:
    zero, add, sub, mul, inv = 0, staticmethod(lambda a, b: (a + b) % P), staticmethod(lambda a, b: (a - b) % P), \
        staticmethod(lambda a, b: a * b % P), staticmethod(_inv)
    small = staticmethod(lambda k: k % P)


class _Ops2:
    zero, add, sub, mul, inv = (0, 0), staticmethod(f2_add), staticmethod(f2_sub), staticmethod(f2_mul), staticmethod(f2_inv)
    small = staticmethod(lambda k: (k % P, 0))


def _pt_add(o, p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if o.add(p[1], q[1]) == o.zero:
            return None
        lam = o.mul(o.mul(o.small(3), o.mul(p[0], p[0])), o.inv(o.mul(o.small(2), p[1])))
    else:
        lam = o.mul(o.sub(q[1], p[1]), o.inv(o.sub(q[0], p[0])))
    x = o.sub(o.sub(o.mul(lam, lam), p[0]), q[0])
    return (x, o.sub(o.mul(lam, o.sub(p[0], x)), p[1]))


def _jac_dbl(o, p):
    X, Y, Z = p
    A = o.mul(X, X); B = o.mul(Y, Y); C = o.mul(B, B)
    t = o.add(X, B)
    D = o.sub(o.sub(o.mul(t, t), A), C); D = o.add(D, D)
    E = o.add(o.add(A, A), A)
    X3 = o.sub(o.mul(E, E), o.add(D, D))
    C8 = o.add(C, C); C8 = o.add(C8, C8); C8 = o.add(C8, C8)
    Y3 = o.sub(o.mul(E, o.sub(D, X3)), C8)
    Z3 = o.mul(Y, Z); Z3 = o.add(Z3, Z3)
    return (X3, Y3, Z3)


def _jac_madd(o, p, q):   # Jacobian p + affine q (p not infinity, q not infinity)
    X1, Y1, Z1 = p
    Z1Z1 = o.mul(Z1, Z1)
    U2 = o.mul(q[0], Z1Z1)
    S2 = o.mul(o.mul(q[1], Z1), Z1Z1)
    if U2 == X1:
        if S2 == Y1:
            return _jac_dbl(o, p)
        return None
    H = o.sub(U2, X1)
    HH = o.mul(H, H)
    I = o.add(HH, HH); I = o.add(I, I)
    J = o.mul(H, I)
    rr = o.sub(S2, Y1); rr = o.add(rr, rr)
    V = o.mul(X1, I)
    X3 = o.sub(o.sub(o.mul(rr, rr), J), o.add(V, V))
    YJ = o.mul(Y1, J)
    Y3 = o.sub(o.mul(rr, o.sub(V, X3)), o.add(YJ, YJ))
    ZH = o.add(Z1, H)
    Z3 = o.sub(o.sub(o.mul(ZH, ZH), Z1Z1), HH)
    return (X3, Y3, Z3)


def _pt_mul(o, p, k):
    """k * p, Jacobian double-and-add (one inversion at the end)"""
    k %= R
    if p is None or k == 0:
        return None
    acc = None
    one = o.small(1)
    for i in range(k.bit_length() - 1, -1, -1):
        if acc is not None:
            acc = _jac_dbl(o, acc)
        if (k >> i) & 1:
            acc = (p[0], p[1], one) if acc is None else _jac_madd(o, acc, p)
    if acc is None or acc[2] == o.zero:
        return None
    zi = o.inv(acc[2])
    zi2 = o.mul(zi, zi)
    return (o.mul(acc[0], zi2), o.mul(acc[1], o.mul(zi2, zi)))


def g1_mul(k): return _pt_mul(_Ops1, G1, k)
def g2_mul(k): return _pt_mul(_Ops2, G2, k)
def g1_on_curve(p): return (p[1] * p[1] - p[0] ** 3 - 3) % P == 0


def g2_on_curve(p):
    b = f2_mul((3, 0), f2_inv((9, 1)))  # y^2 = x^3 + 3/(9+u)
    return f2_sub(f2_mul(p[1], p[1]), f2_add(f2_mul(p[0], f2_mul(p[0], p[0])), b)) == (0, 0)
