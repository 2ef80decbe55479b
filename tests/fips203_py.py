"""An independent, deliberately naive pure-Python restatement of FIPS 203 (ML-KEM) — test infrastructure only.

Written from the standard's algorithm listings (Algorithms 3-21) with hashlib for SHA-3/SHAKE; it shares no code
with oracle/mlkem_oracle.c.  Its purpose is to pin the oracle's (and thereby the engine's) FIPS-203 conformance mode,
for which the reference repository has no vectors: PRF = SHAKE256, J = SHAKE256, H = SHA3-256, G = SHA3-512."""
import hashlib

Q, N = 3329, 256
PARAMS = {512: (2, 3, 2, 10, 4), 768: (3, 2, 2, 10, 4), 1024: (4, 2, 2, 11, 5)}


def bitrev7(i):
    return int(format(i, "07b")[::-1], 2)


ZETA = [pow(17, bitrev7(i), Q) for i in range(128)]
GAMMA = [pow(17, 2 * bitrev7(i) + 1, Q) for i in range(128)]


def bytes_to_bits(B):
    return [(b >> j) & 1 for b in B for j in range(8)]


def bits_to_bytes(bits):
    return bytes(sum(bits[8 * i + j] << j for j in range(8)) for i in range(len(bits) // 8))


def byte_encode(F, d):
    bits = []
    for a in F:
        bits += [(a >> j) & 1 for j in range(d)]
    return bits_to_bytes(bits)


def byte_decode(B, d):
    bits = bytes_to_bits(B)
    m = (1 << d) if d < 12 else Q
    return [sum(bits[i * d + j] << j for j in range(d)) % m for i in range(N)]


def compress(x, d):
    return ((x << d) + Q // 2) // Q % (1 << d)


def decompress(y, d):
    return (y * Q + (1 << (d - 1))) >> d


def sample_ntt(B):
    stream = hashlib.shake_128(bytes(B)).digest(3 * 1000)
    a, i = [], 0
    while len(a) < N:
        c0, c1, c2 = stream[i], stream[i + 1], stream[i + 2]
        d1, d2 = c0 + 256 * (c1 % 16), c1 // 16 + 16 * c2
        if d1 < Q:
            a.append(d1)
        if d2 < Q and len(a) < N:
            a.append(d2)
        i += 3
    return a


def sample_cbd(B, eta):
    b = bytes_to_bits(B)
    return [(sum(b[2 * i * eta + j] for j in range(eta)) - sum(b[2 * i * eta + eta + j] for j in range(eta))) % Q for i in range(N)]


def ntt(f):
    f = list(f)
    i, ln = 1, 128
    while ln >= 2:
        for start in range(0, N, 2 * ln):
            z = ZETA[i]
            i += 1
            for j in range(start, start + ln):
                t = z * f[j + ln] % Q
                f[j + ln] = (f[j] - t) % Q
                f[j] = (f[j] + t) % Q
        ln //= 2
    return f


def intt(f):
    f = list(f)
    i, ln = 127, 2
    while ln <= 128:
        for start in range(0, N, 2 * ln):
            z = ZETA[i]
            i -= 1
            for j in range(start, start + ln):
                t = f[j]
                f[j] = (t + f[j + ln]) % Q
                f[j + ln] = z * (f[j + ln] - t) % Q
        ln *= 2
    return [x * 3303 % Q for x in f]


def multiply_ntts(f, g):
    h = [0] * N
    for i in range(128):
        a0, a1, b0, b1 = f[2 * i], f[2 * i + 1], g[2 * i], g[2 * i + 1]
        h[2 * i] = (a0 * b0 + a1 * b1 * GAMMA[i]) % Q
        h[2 * i + 1] = (a0 * b1 + a1 * b0) % Q
    return h


def padd(a, b):
    return [(x + y) % Q for x, y in zip(a, b)]


def prf(eta, s, b):
    return hashlib.shake_256(bytes(s) + bytes([b])).digest(64 * eta)


def H(x):
    return hashlib.sha3_256(bytes(x)).digest()


def G(x):
    return hashlib.sha3_512(bytes(x)).digest()


def J(x):
    return hashlib.shake_256(bytes(x)).digest(32)


def pke_keygen(pset, d):
    k, eta1, _, _, _ = PARAMS[pset]
    g = G(bytes(d) + bytes([k]))
    rho, sigma = g[:32], g[32:]
    A = [[sample_ntt(rho + bytes([j, i])) for j in range(k)] for i in range(k)]
    s = [ntt(sample_cbd(prf(eta1, sigma, n), eta1)) for n in range(k)]
    e = [ntt(sample_cbd(prf(eta1, sigma, k + n), eta1)) for n in range(k)]
    t = []
    for i in range(k):
        acc = [0] * N
        for j in range(k):
            acc = padd(acc, multiply_ntts(A[i][j], s[j]))
        t.append(padd(acc, e[i]))
    return b"".join(byte_encode(x, 12) for x in t) + rho, b"".join(byte_encode(x, 12) for x in s)


def pke_encrypt(pset, ek, m, r):
    k, eta1, eta2, du, dv = PARAMS[pset]
    t = [byte_decode(ek[384 * i:384 * i + 384], 12) for i in range(k)]
    rho = ek[384 * k:]
    A = [[sample_ntt(rho + bytes([j, i])) for j in range(k)] for i in range(k)]
    y = [ntt(sample_cbd(prf(eta1, r, n), eta1)) for n in range(k)]
    e1 = [sample_cbd(prf(eta2, r, k + n), eta2) for n in range(k)]
    e2 = sample_cbd(prf(eta2, r, 2 * k), eta2)
    u = []
    for i in range(k):
        acc = [0] * N
        for j in range(k):
            acc = padd(acc, multiply_ntts(A[j][i], y[j]))   # A^T
        u.append(padd(intt(acc), e1[i]))
    mu = [decompress(b, 1) for b in byte_decode(m, 1)]
    acc = [0] * N
    for j in range(k):
        acc = padd(acc, multiply_ntts(t[j], y[j]))
    v = padd(padd(intt(acc), e2), mu)
    return b"".join(byte_encode([compress(x, du) for x in ui], du) for ui in u) + byte_encode([compress(x, dv) for x in v], dv)


def pke_decrypt(pset, dk, c):
    k, _, _, du, dv = PARAMS[pset]
    u = [[decompress(x, du) for x in byte_decode(c[32 * du * i:32 * du * (i + 1)], du)] for i in range(k)]
    v = [decompress(x, dv) for x in byte_decode(c[32 * du * k:], dv)]
    s = [byte_decode(dk[384 * i:384 * i + 384], 12) for i in range(k)]
    acc = [0] * N
    for j in range(k):
        acc = padd(acc, multiply_ntts(s[j], ntt(u[j])))
    w = [(a - b) % Q for a, b in zip(v, intt(acc))]
    return byte_encode([compress(x, 1) for x in w], 1)


def keygen(pset, d, z):
    ek, dk_pke = pke_keygen(pset, d)
    return ek, dk_pke + ek + H(ek) + bytes(z)


def encaps(pset, ek, m):
    g = G(bytes(m) + H(ek))
    return pke_encrypt(pset, ek, bytes(m), g[32:]), g[:32]


def decaps(pset, dk, c):
    k = PARAMS[pset][0]
    dk_pke, ek, h, z = dk[:384 * k], dk[384 * k:768 * k + 32], dk[768 * k + 32:768 * k + 64], dk[768 * k + 64:]
    m2 = pke_decrypt(pset, dk_pke, c)
    g = G(m2 + h)
    kbar = J(z + bytes(c))
    return g[:32] if pke_encrypt(pset, ek, m2, g[32:]) == bytes(c) else kbar


def modulus_check(pset, ek):
    k = PARAMS[pset][0]
    return all(byte_encode(byte_decode(ek[384 * i:384 * i + 384], 12), 12) == ek[384 * i:384 * i + 384] for i in range(k))
