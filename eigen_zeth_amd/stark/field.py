"""Scalar Goldilocks / F_{p^3} helpers for host-side protocol bookkeeping (challenges, a handful of
constants per proof).  Python ints; nothing here touches O(trace) data -- that is the kernels' job."""
P = 0xFFFFFFFF00000001


def root(logn, root32):
    return pow(root32, 1 << (32 - logn), P)


def inv(a):
    return pow(a, P - 2, P)


def e3(a):
    return [int(a[0]) % P, int(a[1]) % P, int(a[2]) % P]


def e3_add(a, b):
    return [(a[i] + b[i]) % P for i in range(3)]


def e3_sub(a, b):
    return [(a[i] - b[i]) % P for i in range(3)]


def e3_mul(a, b):
    d = [0] * 5
    for i in range(3):
        for j in range(3):
            d[i + j] += a[i] * b[j]
    return [(d[0] + d[3]) % P, (d[1] + d[3] + d[4]) % P, (d[2] + d[4]) % P]  # t^3 = t + 1


def e3_scale(a, s):
    return [(x * s) % P for x in a]


def e3_pow(a, e):
    r = [1, 0, 0]
    while e:
        if e & 1:
            r = e3_mul(r, a)
        a = e3_mul(a, a)
        e >>= 1
    return r


def e3_inv(a):
    return e3_pow(a, P ** 3 - 2)


def embed(v):
    return [int(v) % P, 0, 0]
