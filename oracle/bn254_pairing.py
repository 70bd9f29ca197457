"""oracle/bn254_pairing.py -- optimal-ate pairing on alt_bn128 in pure Python (TEST INFRASTRUCTURE).
Textbook construction: F_q^12 = F_q[w]/(w^12 - 18 w^6 + 82), G2 mapped into E(F_q^12) by the twist
(x,y) -> (x' w^2, y' w^3) with x' = (c0 - 9 c1) + c1 w^6, Miller loop over 29793968203157093288 with
the two Frobenius line steps, final exponent (q^12 - 1)/r.  Sanity: bilinearity and non-degeneracy
(tests/test_groth16.py).  Used only to verify the Groth16 proofs the product emits."""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE = 29793968203157093288
LOG_ATE = 63


def f_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for i in range(22, 11, -1):
        top = t[i]
        if top:
            t[i - 12] -= 82 * top
            t[i - 6] += 18 * top
    return [v % Q for v in t[:12]]


def f_add(a, b): return [(x + y) % Q for x, y in zip(a, b)]
def f_sub(a, b): return [(x - y) % Q for x, y in zip(a, b)]
def f_scalar(a, k): return [x * k % Q for x in a]
ONE = [1] + [0] * 11
ZERO = [0] * 12


def _deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def _poly_div(a, b):
    dega, degb = _deg(a), _deg(b)
    temp, o = list(a), [0] * len(a)
    binv = pow(b[degb], Q - 2, Q)
    for i in range(dega - degb, -1, -1):
        o[i] = (o[i] + temp[degb + i] * binv) % Q
        for c in range(degb + 1):
            temp[c + i] = (temp[c + i] - o[i] * b[c]) % Q
    return o[:_deg(o) + 1]


def f_inv(a):
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % Q, 0, 0, 0, 0, 0, 1]
    while _deg(low):
        r = _poly_div(high, low)
        r += [0] * (13 - len(r))
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                new[i + j] = (new[i + j] - low[i] * r[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], Q - 2, Q)
    return [v * inv0 % Q for v in lm[:12]]


def f_pow(a, e):
    r = ONE
    while e:
        if e & 1:
            r = f_mul(r, a)
        a = f_mul(a, a)
        e >>= 1
    return r


def embed1(p):   # G1 point -> E(F_q^12)
    return ([p[0]] + [0] * 11, [p[1]] + [0] * 11)


def twist(p):    # G2 point ((x0,x1),(y0,y1)) -> E(F_q^12)
    (x0, x1), (y0, y1) = p
    nx = [(x0 - 9 * x1) % Q] + [0] * 5 + [x1] + [0] * 5
    ny = [(y0 - 9 * y1) % Q] + [0] * 5 + [y1] + [0] * 5
    w2 = [0, 0, 1] + [0] * 9
    w3 = [0, 0, 0, 1] + [0] * 8
    return (f_mul(nx, w2), f_mul(ny, w3))


def _dbl(p):
    x, y = p
    m = f_mul(f_scalar(f_mul(x, x), 3), f_inv(f_scalar(y, 2)))
    nx = f_sub(f_mul(m, m), f_scalar(x, 2))
    return (nx, f_sub(f_mul(m, f_sub(x, nx)), y))


def _add(p, q):
    if p[0] == q[0]:
        return _dbl(p) if p[1] == q[1] else None
    m = f_mul(f_sub(q[1], p[1]), f_inv(f_sub(q[0], p[0])))
    nx = f_sub(f_sub(f_mul(m, m), p[0]), q[0])
    return (nx, f_sub(f_mul(m, f_sub(p[0], nx)), p[1]))


def _line(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f_mul(f_sub(y2, y1), f_inv(f_sub(x2, x1)))
    elif y1 == y2:
        m = f_mul(f_scalar(f_mul(x1, x1), 3), f_inv(f_scalar(y1, 2)))
    else:
        return f_sub(xt, x1)
    return f_sub(f_mul(m, f_sub(xt, x1)), f_sub(yt, y1))


def miller(q2, p1):
    """Miller loop value (no final exponentiation); q2 a G2 point, p1 a G1 point"""
    if q2 is None or p1 is None:
        return ONE
    Qp, Pp = twist(q2), embed1(p1)
    Rp, f = Qp, ONE
    for i in range(LOG_ATE, -1, -1):
        f = f_mul(f_mul(f, f), _line(Rp, Rp, Pp))
        Rp = _dbl(Rp)
        if ATE & (1 << i):
            f = f_mul(f, _line(Rp, Qp, Pp))
            Rp = _add(Rp, Qp)
    Q1 = (f_pow(Qp[0], Q), f_pow(Qp[1], Q))
    nQ2 = (f_pow(Q1[0], Q), [(-v) % Q for v in f_pow(Q1[1], Q)])
    f = f_mul(f, _line(Rp, Q1, Pp))
    Rp = _add(Rp, Q1)
    f = f_mul(f, _line(Rp, nQ2, Pp))
    return f


def final_exp(f):
    return f_pow(f, (Q ** 12 - 1) // R)


def pairing(q2, p1):
    return final_exp(miller(q2, p1))
