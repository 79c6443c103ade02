"""ORACLE (test infrastructure, never shipped, never on the product path).

Pure-Python big-integer restatement of the group/field layer the reference
gets from `curve25519-dalek = "2"` (reference Cargo.toml:8; not on disk).
Restated from RFC 9496 (ristretto255) and the published dalek algorithms.

Pinned by tests/test_oracle_kats.py against RFC 9496 vectors (basepoint
multiples, hash-to-group) and the documented bulletproofs `B_blinding`.

  Scalar      <-> curve25519_dalek::scalar::Scalar        
  Point       <-> RistrettoPoint / CompressedRistretto    (SURVEY §8a P1,P2)
"""

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def _is_neg(x):
    return (x % P) & 1


def _abs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 §4.2 SQRT_RATIO_M1 -> (was_square, r)."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (P - u) % P
    flipped_i = check == (P - u) * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    return (correct or flipped), _abs(r)


ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P
_, INVSQRT_A_MINUS_D = sqrt_ratio_m1(1, (-1 - D) % P)
# RFC 9496 fixes the root of a*d-1 explicitly (the odd-looking one).
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
assert SQRT_AD_MINUS_ONE * SQRT_AD_MINUS_ONE % P == (-D - 1) % P


class Point:
    """Extended twisted-Edwards coordinates (X:Y:Z:T), a=-1."""
    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % P, Y % P, Z % P, T % P

    @staticmethod
    def identity():
        return Point(0, 1, 1, 0)

    def __add__(self, o):
        # add-2008-hwcd-3 (a=-1), as in dalek's ProjectiveNiels addition.
        A = (self.Y - self.X) * (o.Y - o.X) % P
        B = (self.Y + self.X) * (o.Y + o.X) % P
        C = self.T * 2 * D % P * o.T % P
        Dd = self.Z * 2 * o.Z % P
        E, F, G, H = B - A, Dd - C, Dd + C, B + A
        return Point(E * F, G * H, F * G, E * H)

    def double(self):
        A = self.X * self.X % P
        B = self.Y * self.Y % P
        C = 2 * self.Z * self.Z % P
        H = A + B
        E = H - (self.X + self.Y) ** 2 % P
        G = A - B
        F = C + G
        return Point(E * F, G * H, F * G, E * H)

    def __neg__(self):
        return Point(-self.X, self.Y, self.Z, -self.T)

    def __sub__(self, o):
        return self + (-o)

    def __mul__(self, k):
        k %= L
        acc, base = Point.identity(), self
        while k:
            if k & 1:
                acc = acc + base
            base = base.double()
            k >>= 1
        return acc

    __rmul__ = __mul__

    def __eq__(self, o):
        # ristretto equality: X1*Y2 == Y1*X2  or  Y1*Y2 == X1*X2
        return (self.X * o.Y - self.Y * o.X) % P == 0 or (self.Y * o.Y - self.X * o.X) % P == 0

    def is_identity(self):
        return self == Point.identity()

    def compress(self):
        """RFC 9496 §4.3.2 Encode."""
        X, Y, Z, T = self.X, self.Y, self.Z, self.T
        u1 = (Z + Y) * (Z - Y) % P
        u2 = X * Y % P
        _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
        den1 = invsqrt * u1 % P
        den2 = invsqrt * u2 % P
        z_inv = den1 * den2 % P * T % P
        ix0 = X * SQRT_M1 % P
        iy0 = Y * SQRT_M1 % P
        enchanted = den1 * INVSQRT_A_MINUS_D % P
        rotate = _is_neg(T * z_inv)
        if rotate:
            x, y, den_inv = iy0, ix0, enchanted
        else:
            x, y, den_inv = X, Y, den2
        if _is_neg(x * z_inv):
            y = (-y) % P
        s = _abs(den_inv * ((Z - y) % P))
        return s.to_bytes(32, "little")


def decompress(b):
    """RFC 9496 §4.3.1 Decode -> Point or None."""
    assert len(b) == 32
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = _abs(2 * s * den_x)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or _is_neg(t) or y == 0:
        return None
    return Point(x, y, 1, t)


def _elligator(r0):
    """RFC 9496 §4.3.4 MAP."""
    r = SQRT_M1 * r0 % P * r0 % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-_abs(s * r0)) % P
    if not was_square:
        s = s_prime
        c = r
    else:
        c = P - 1
    N = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - v) % P
    ss = s * s % P
    w0 = 2 * s * v % P
    w1 = N * SQRT_AD_MINUS_ONE % P
    w2 = (1 - ss) % P
    w3 = (1 + ss) % P
    return Point(w0 * w3, w2 * w1, w1 * w3, w0 * w2)


def from_uniform_bytes(b):
    """RistrettoPoint::from_uniform_bytes (RFC 9496 §4.3.4)."""
    assert len(b) == 64
    r1 = int.from_bytes(b[:32], "little") & (2**255 - 1)
    r2 = int.from_bytes(b[32:], "little") & (2**255 - 1)
    return _elligator(r1 % P) + _elligator(r2 % P)


# Ed25519 basepoint == ristretto255 generator
_by = 4 * pow(5, P - 2, P) % P
_bx2 = (_by * _by - 1) * pow(D * _by * _by + 1, P - 2, P) % P
_ok, _bx = sqrt_ratio_m1(_bx2, 1)
assert _ok
if _bx & 1:
    _bx = P - _bx
BASEPOINT = Point(_bx, _by, 1, _bx * _by)


# ---- scalar helpers (mod l); scalars are plain ints in [0, L) -------------
def sc_from_bytes_mod_order(b):
    return int.from_bytes(b, "little") % L


def sc_from_bytes_wide(b):
    assert len(b) == 64
    return int.from_bytes(b, "little") % L


def sc_to_bytes(s):
    return (s % L).to_bytes(32, "little")


def sc_invert(s):
    """Scalar::invert — Fermat; 0 -> 0 like dalek."""
    return pow(s % L, L - 2, L)


def msm(scalars, points):
    """Plain multiscalar multiplication (value-equivalent to dalek's
    Straus/Pippenger; group arithmetic is exact)."""
    scalars = [s % L for s in scalars]
    # simple 4-bit fixed-window Straus
    tables = []
    for p in points:
        t = [Point.identity(), p]
        for _ in range(14):
            t.append(t[-1] + p)
        tables.append(t)
    acc = Point.identity()
    for w in range(63, -1, -1):
        if w != 63:
            acc = acc.double().double().double().double()
        for s, t in zip(scalars, tables):
            d = (s >> (4 * w)) & 15
            if d:
                acc = acc + t[d]
    return acc
