"""oracle/naive_bn254.py -- definition-level BN254 G1 arithmetic and MSM on Python ints (TEST
INFRASTRUCTURE).  Affine addition with modular inverses, double-and-add; no shared code with the HIP
kernel (Jacobian/Montgomery) or with the service's point encoder.  PARITY UNPINNED w.r.t. the
external prover; the curve itself is the public alt_bn128 (y^2 = x^3 + 3, generator (1,2))."""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
G = (1, 2)


def add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], Q - 2, Q) % Q
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], Q - 2, Q) % Q
    x = (lam * lam - p[0] - q[0]) % Q
    return (x, (lam * (p[0] - x) - p[1]) % Q)


def mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = add(acc, p)
        p = add(p, p)
        k >>= 1
    return acc


def msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        if p is None or p == (0, 0):
            continue
        acc = add(acc, mul(p, s % R))
    return acc


def on_curve(p):
    return (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0


# ---- G2: the sextic twist y^2 = x^3 + 3/(9+u) over F_q2 = F_q[u]/(u^2+1); elements are (c0, c1) tuples.
# Affine chord-and-tangent with explicit F_q2 inversion -- nothing shared with the HIP kernel (Jacobian,
# Karatsuba on 29-bit Montgomery limbs).
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], Q - 2, Q)
    return (a[0] * d % Q, -a[1] * d % Q)


B2 = f2_mul((3, 0), f2_inv((9, 1)))


def on_curve_g2(p):
    return f2_sub(f2_mul(p[1], p[1]), f2_add(f2_mul(f2_mul(p[0], p[0]), p[0]), B2)) == (0, 0)


def add_g2(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if f2_add(p[1], q[1]) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(p[0], p[0])), f2_inv(f2_add(p[1], p[1])))
    else:
        lam = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x = f2_sub(f2_sub(f2_mul(lam, lam), p[0]), q[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(p[0], x)), p[1]))


def mul_g2(p, k):
    acc = None
    while k:
        if k & 1:
            acc = add_g2(acc, p)
        p = add_g2(p, p)
        k >>= 1
    return acc


def msm_g2(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        if p is None or p == ((0, 0), (0, 0)):
            continue
        acc = add_g2(acc, mul_g2(p, s % R))
    return acc
