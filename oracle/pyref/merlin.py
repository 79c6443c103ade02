"""ORACLE (test infrastructure, never shipped, never on the product path).

Keccak-f[1600], SHAKE256, SHA3-512, STROBE-128 and the Merlin transcript /
TranscriptRng, restated from the published specs (FIPS 202, STROBE v1.0.2,
merlin 2.0 `strobe.rs` / `transcript.rs`).  The reference binds to this
behaviour via `merlin = "2"` (Cargo.toml:18) and `Transcript::new(label)` at
e.g. src/gadget_vsmt_4.rs:390, src/gadget_bound_check.rs:58.

Pinned by merlin's published equivalence test vector
(tests/test_oracle_kats.py::test_merlin_kat).
"""
import struct

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61],
        [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]  # _ROT[x][y]
_M = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M if n else v


def keccak_f1600(lanes):
    """lanes: list of 25 u64, index x+5y. In-place."""
    A = lanes
    for rc in _RC:
        C = [A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20] for x in range(5)]
        Dd = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        for i in range(25):
            A[i] ^= Dd[i % 5]
        B = [0] * 25
        for x in range(5):
            for y in range(5):
                B[y + 5 * ((2 * x + 3 * y) % 5)] = _rol(A[x + 5 * y], _ROT[x][y])
        for y in range(5):
            for x in range(5):
                A[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & _M & B[(x + 2) % 5 + 5 * y])
        A[0] ^= rc
    return A


def _permute_bytes(state):
    lanes = list(struct.unpack("<25Q", bytes(state)))
    keccak_f1600(lanes)
    state[:] = struct.pack("<25Q", *lanes)


def _sponge(rate, suffix, data, outlen):
    st = bytearray(200)
    data = bytearray(data) + bytes([suffix])
    while len(data) % rate:
        data.append(0)
    data[-1] |= 0x80
    for off in range(0, len(data), rate):
        for i in range(rate):
            st[i] ^= data[off + i]
        _permute_bytes(st)
    out = bytearray()
    while len(out) < outlen:
        out += st[:rate]
        if len(out) < outlen:
            _permute_bytes(st)
    return bytes(out[:outlen])


def shake256(data, outlen):
    return _sponge(136, 0x1F, data, outlen)


def sha3_512(data):
    return _sponge(72, 0x06, data, 64)


# ---------------------------------------------------------------- STROBE-128
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32
STROBE_R = 166


class Strobe128:
    def __init__(self, protocol_label=None):
        if protocol_label is None:
            return
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        _permute_bytes(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def clone(self):
        c = Strobe128()
        c.state, c.pos, c.pos_begin, c.cur_flags = bytearray(self.state), self.pos, self.pos_begin, self.cur_flags
        return c

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        _permute_bytes(self.state)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.state[self.pos] ^= b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _overwrite(self, data):
        for b in data:
            self.state[self.pos] = b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert not (flags & FLAG_T)
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if (flags & (FLAG_C | FLAG_K)) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more=False):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)

    def key(self, data, more=False):
        self._begin_op(FLAG_A | FLAG_C, more)
        self._overwrite(data)


def _le32(n):
    return struct.pack("<I", n)


class Transcript:
    """merlin::Transcript plus the bulletproofs TranscriptProtocol helpers."""

    def __init__(self, label=None):
        if label is None:
            return
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def clone(self):
        t = Transcript()
        t.strobe = self.strobe.clone()
        return t

    def append_message(self, label, msg):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_le32(len(msg)), True)
        self.strobe.ad(msg, False)

    def append_u64(self, label, v):
        self.append_message(label, struct.pack("<Q", v))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_le32(n), True)
        return self.strobe.prf(n)

    # --- bulletproofs TranscriptProtocol (dalek bulletproofs transcript.rs) ---
    def challenge_scalar(self, label):
        from .ed import sc_from_bytes_wide
        return sc_from_bytes_wide(self.challenge_bytes(label, 64))

    def append_scalar(self, label, s):
        from .ed import sc_to_bytes
        self.append_message(label, sc_to_bytes(s))

    def append_point(self, label, compressed32):
        self.append_message(label, compressed32)

    def validate_and_append_point(self, label, compressed32):
        if compressed32 == bytes(32):
            raise VerificationError("identity point")
        self.append_message(label, compressed32)

    def build_rng(self):
        return TranscriptRngBuilder(self.strobe.clone())


class TranscriptRngBuilder:
    def __init__(self, strobe):
        self.strobe = strobe

    def rekey_with_witness_bytes(self, label, witness):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(_le32(len(witness)), True)
        self.strobe.key(witness)
        return self

    def finalize(self, external_32_bytes):
        """Upstream draws these 32 bytes from `thread_rng()`; the oracle
        makes them an explicit input (SURVEY §8c determinism convention)."""
        assert len(external_32_bytes) == 32
        self.strobe.meta_ad(b"rng", False)
        self.strobe.key(external_32_bytes)
        return TranscriptRng(self.strobe)


class TranscriptRng:
    def __init__(self, strobe):
        self.strobe = strobe

    def fill_bytes(self, n):
        self.strobe.meta_ad(_le32(n), False)
        return self.strobe.prf(n)

    def random_scalar(self):
        """Scalar::random(rng): 64 bytes -> from_bytes_mod_order_wide."""
        from .ed import sc_from_bytes_wide
        return sc_from_bytes_wide(self.fill_bytes(64))


class VerificationError(Exception):
    pass
