"""Device arithmetic headers (field, scalars, ristretto, Keccak/STROBE/Merlin) compiled for the
host and compared with the oracle's big-integer arithmetic."""
import ctypes
import random

from pyref.ed import P, L, BASEPOINT, from_uniform_bytes
from pyref.merlin import Transcript
from pyref.ed import sc_to_bytes


def _call(f, *ins, n=32):
    o = ctypes.create_string_buffer(n)
    r = f(*ins, o)
    return o.raw, r


def _vals():
    rnd = random.Random(7)
    edge = [0, 1, 2, 19, 38, P - 1, P, P + 1, 2**255 - 1, 2**255, 2**256 - 1, 2**256 - 38, 2**256 - 39, L - 1, L, L + 1, 2**252]
    return [x.to_bytes(32, "little") for x in edge] + [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(120)]


def test_field_and_scalar_arithmetic(prim_lib):
    vals = _vals()
    for a in vals:
        ia = int.from_bytes(a, "little")
        for b in vals[:20]:
            ib = int.from_bytes(b, "little")
            for f, op, mod in ((prim_lib.hs_fe_mul, ia * ib, P), (prim_lib.hs_fe_add, ia + ib, P), (prim_lib.hs_fe_sub, ia - ib, P),
                               (prim_lib.hs_sc_mul, ia * ib, L), (prim_lib.hs_sc_add, ia + ib, L), (prim_lib.hs_sc_sub, ia - ib, L)):
                assert int.from_bytes(_call(f, a, b)[0], "little") == op % mod
        assert int.from_bytes(_call(prim_lib.hs_fe_inv, a)[0], "little") == pow(ia % P, P - 2, P)
        assert int.from_bytes(_call(prim_lib.hs_fe_sq, a)[0], "little") == ia * ia % P
        inv = pow(ia % L, L - 2, L)
        assert int.from_bytes(_call(prim_lib.hs_sc_inv, a)[0], "little") == inv          # safegcd divsteps
        assert int.from_bytes(_call(prim_lib.hs_sc_inv_fermat, a)[0], "little") == inv   # Fermat ladder
        assert int.from_bytes(_call(prim_lib.hs_sc_inv_var, a)[0], "little") == inv      # variable-time divsteps (public values)


def test_variable_time_inverse_on_many_values(prim_lib):
    rnd = random.Random(77)
    vals = [1 << k for k in range(0, 253, 7)] + [L - (1 << k) for k in range(0, 250, 11)] + [(1 << k) - 1 for k in range(1, 253, 9)]
    vals += [rnd.getrandbits(rnd.choice((8, 31, 60, 61, 120, 200, 252))) for _ in range(1500)]
    for x in vals:
        x %= L
        got = int.from_bytes(_call(prim_lib.hs_sc_inv_var, x.to_bytes(32, "little"))[0], "little")
        assert got == (pow(x, L - 2, L) if x else 0), hex(x)


def test_group_and_encoding(prim_lib):
    rnd = random.Random(9)
    for _ in range(12):
        w = bytes(rnd.getrandbits(8) for _ in range(64))
        assert int.from_bytes(_call(prim_lib.hs_sc_wide, w)[0], "little") == int.from_bytes(w, "little") % L
        assert _call(prim_lib.hs_uniform, w)[0] == from_uniform_bytes(w).compress()
        k = bytes(rnd.getrandbits(8) for _ in range(32))
        p = BASEPOINT * int.from_bytes(k, "little")
        assert _call(prim_lib.hs_basemul, k)[0] == p.compress()
        out, ok = _call(prim_lib.hs_decompress_recompress, p.compress())
        assert ok and out == p.compress()
        q = from_uniform_bytes(bytes(rnd.getrandbits(8) for _ in range(64)))
        o4, ok = _call(prim_lib.hs_addsub, p.compress(), q.compress(), n=128)
        assert ok and o4 == (p + q).compress() + (p - q).compress() + (p + q).compress() + (p - q).compress()
    assert _call(prim_lib.hs_decompress_recompress, b"\x01" + bytes(31))[1] == 0


def test_merlin_transcript_and_rng(prim_lib):
    assert _call(prim_lib.hs_merlin_kat)[0].hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    rnd = random.Random(11)
    msgs = bytes(rnd.getrandbits(8) for _ in range(5 * 32))
    seed = bytes(rnd.getrandbits(8) for _ in range(32))
    dr, ch = ctypes.create_string_buffer(32 * 300), ctypes.create_string_buffer(32)
    prim_lib.hs_merlin_script(b"VSMT", 4, msgs, 5, seed, 300, dr, ch)
    t = Transcript(b"VSMT")
    for i in range(5):
        t.append_message(b"V", msgs[32 * i:32 * i + 32])
    t.append_u64(b"m", 5)
    b = t.build_rng()
    for i in range(5):
        b = b.rekey_with_witness_bytes(b"v_blinding", msgs[32 * i:32 * i + 32])
    rng = b.finalize(seed)
    assert dr.raw == b"".join(sc_to_bytes(rng.random_scalar()) for _ in range(300))
    assert ch.raw == sc_to_bytes(t.challenge_scalar(b"y"))


def test_field_limb_bounds(prim_lib):
    """9x29 signed-limb field (csrc/fe.hpp): worst-case limb classes of every product shape used in ge.hpp
    (N*N, 2N*2N, 2N*3N, 3N*3N, 3N*4N, 2N*4N; squares up to 2N) stay exact and return limbs within N."""
    import ctypes, random
    P = 2**255 - 19
    N = 2**28 + 2**23
    I9 = ctypes.c_int32 * 9
    rnd = random.Random(29)

    def val(l):
        return sum(int(x) << (29 * i) for i, x in enumerate(l)) % P

    def patterns(bound):
        yield [bound] * 9
        yield [-bound] * 9
        yield [bound if i % 2 else -bound for i in range(9)]
        yield [-bound if i % 2 else bound for i in range(9)]
        for _ in range(40):
            yield [rnd.choice((bound, -bound, rnd.randint(-bound, bound))) for _ in range(9)]

    out = (ctypes.c_uint8 * 32)()
    ol = I9()
    for ka, kb in ((1, 1), (2, 2), (2, 3), (3, 3), (3, 4), (2, 4), (4, 1)):
        for a in patterns(ka * N):
            for b in list(patterns(kb * N))[:12]:
                prim_lib.hs_fe_mul_limbs(I9(*a), I9(*b), out, ol)
                assert int.from_bytes(bytes(out), "little") == val(a) * val(b) % P
                assert max(abs(x) for x in ol) <= N
    for a in patterns(2 * N):
        prim_lib.hs_fe_sq_limbs(I9(*a), out, ol)
        assert int.from_bytes(bytes(out), "little") == val(a) ** 2 % P
        assert max(abs(x) for x in ol) <= N
    for k in (1, 2, 3, 4, 7):
        for a in patterns(k * N):
            prim_lib.hs_fe_canon_limbs(I9(*a), out)
            assert int.from_bytes(bytes(out), "little") == val(a)
            prim_lib.hs_fe_carry_limbs(I9(*a), ol)
            assert val(list(ol)) == val(a) and max(abs(x) for x in ol) <= N
    # canonical edge values
    for v in (0, 1, 19, P - 1, P, P + 1, 2**255 - 1, 2**255, 2**256 - 1):
        limbs = [(v >> (29 * i)) & (2**29 - 1) for i in range(9)]
        prim_lib.hs_fe_canon_limbs(I9(*limbs), out)
        assert int.from_bytes(bytes(out), "little") == v % P


def test_table_class_limb_bounds(prim_lib):
    """ge_madd_t (csrc/ge.hpp): the table-addition chain keeps X, Y, T as floor-carry products (limbs in [-2^24, F'),
    F' = 2^29 + 2^24) and Z centred; worst-case limb patterns of that class must multiply exactly and return
    limbs of the same class, for both signs of the digit."""
    import ctypes, random
    P = 2**255 - 19
    N = 2**28 + 2**23
    FP = 2**29 + 2**24
    I9, I27, I36 = ctypes.c_int32 * 9, ctypes.c_int32 * 27, ctypes.c_int32 * 36
    rnd = random.Random(31)

    def val(l):
        return sum(int(x) << (29 * i) for i, x in enumerate(l)) % P

    def floor_pat():
        yield [FP - 1] * 9
        yield [-2**24] * 9
        yield [FP - 1 if i % 2 else -2**24 for i in range(9)]
        for _ in range(6):
            yield [rnd.choice((FP - 1, -2**24, 0, rnd.randint(0, 2**29 - 1))) for _ in range(9)]

    def cent_pat():
        yield [N] * 9
        yield [-N] * 9
        yield [N if i % 2 else -N for i in range(9)]
        for _ in range(4):
            yield [rnd.choice((N, -N, rnd.randint(-N, N))) for _ in range(9)]

    def tab_pat():
        yield [2**29 - 1] * 9
        yield [0] * 9
        for _ in range(3):
            yield [rnd.choice((2**29 - 1, 0, rnd.randint(0, 2**29 - 1))) for _ in range(9)]

    out, ol = (ctypes.c_uint8 * 32)(), I9()
    # the floor-carry multiplier alone, on the largest operand classes ge_madd_t feeds it
    for a in floor_pat():
        for b in floor_pat():
            a2 = [2 * x for x in a]                      # cY <= 2F'
            b3 = [min(3 * N, max(-3 * N, 3 * x)) for x in b]  # |cZ|, |cT| <= N + F' = 3N in ge_madd_t (T*dxy in floor-carry form)
            prim_lib.hs_fe_mul_f_limbs(I9(*a2), I9(*b3), out, ol)
            assert int.from_bytes(bytes(out), "little") == val(a2) * val(b3) % P
            assert min(ol) >= -2**24 and max(ol) < FP
    o36, ob = I36(), (ctypes.c_uint8 * 128)()
    for X in floor_pat():
        for Y in list(floor_pat())[:4]:
            for Z in list(cent_pat())[:4]:
                for T in list(floor_pat())[:3]:
                    for q in tab_pat():
                        qq = list(q) + list(reversed(q)) + [q[(i * 5) % 9] for i in range(9)]
                        for neg in (0, 1):
                            prim_lib.hs_ge_madd_t_limbs(I36(*(X + Y + Z + T)), I27(*qq), neg, o36, ob)
                            x, y, z, t = val(X), val(Y), val(Z), val(T)
                            ypx, ymx, xy2d = val(qq[:9]), val(qq[9:18]), val(qq[18:])
                            if neg:
                                ypx, ymx, xy2d = ymx, ypx, -xy2d
                            A, B, C, D = (y + x) * ypx, (y - x) * ymx, t * xy2d, z   # halved table form: Z, not 2Z
                            cX, cY, cZ, cT = A - B, A + B, D + C, D - C
                            want = [cX * cT % P, cY * cZ % P, cZ * cT % P, cX * cY % P]
                            got = [int.from_bytes(bytes(ob)[32 * k:32 * k + 32], "little") for k in range(4)]
                            assert got == want
                            lim = list(o36)
                            for k in (0, 1, 3):
                                assert min(lim[9 * k:9 * k + 9]) >= -2**24 and max(lim[9 * k:9 * k + 9]) < FP
                            assert max(abs(v) for v in lim[18:27]) <= N


def test_table_msm_formats_and_windows(prim_lib):
    """Fixed-base tables of several window widths (W = 11: 23 windows, the top window keeps
    its digit) against the oracle's big-integer MSM, with edge scalars (0, 1, l-1, 2^252, 2^252 - 1, all-ones windows)."""
    import ctypes, random
    from pyref.ed import Point
    rnd = random.Random(37)
    pts = [from_uniform_bytes(bytes(rnd.getrandbits(8) for _ in range(64))) for _ in range(6)]
    edge = [0, 1, L - 1, 2**252, 2**252 - 1, L - 2**121, int("1" * 252, 2), (2**252 // 3)]
    sets = [edge[i:i + 6] for i in (0, 2)] + [[rnd.randrange(L) for _ in range(6)] for _ in range(2)]
    out = ctypes.create_string_buffer(32)
    for W in (11, 8, 5, 4, 12, 10):
        for ss in sets:
            ok = prim_lib.hs_table_msm(b"".join(p.compress() for p in pts), b"".join(sc_to_bytes(s) for s in ss), 6, W, out)
            want = Point.identity()
            for p, s in zip(pts, ss):
                want = want + p * s
            assert ok and out.raw == want.compress(), W


def test_vb_win_workgroup_order_is_a_permutation(prim_lib):
    for B, VC in [(64, 1), (64, 16), (128, 4), (1024, 16), (1000, 16), (3, 2), (192, 2)]:
        assert prim_lib.hs_vb_win_index_is_permutation(B, VC) == 1, (B, VC)
