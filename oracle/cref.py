"""ORACLE (test infrastructure): ctypes access to the C restatement (oracle/c/bpr1cs_oracle.c).
Used only by tests/, smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libbpr1cs_oracle.so")
BLOB = os.path.join(_HERE, "..", "tests", "golden", "poseidon_params_ristretto.bin")
VSMT_4, POSEIDON_HASH_2, POSEIDON_HASH_4, BOUND_CHECK, VSMT_2, MIMC, SET_MEMBERSHIP, MIMC_SET_MEMBERSHIP = range(8)
GADGET_IDS = {"vsmt_4": VSMT_4, "poseidon_hash_2": POSEIDON_HASH_2, "poseidon_hash_4": POSEIDON_HASH_4, "bound_check": BOUND_CHECK,
              "vsmt_2": VSMT_2, "mimc": MIMC, "set_membership": SET_MEMBERSHIP, "mimc_set_membership": MIMC_SET_MEMBERSHIP}


def _sc(x):
    return x if isinstance(x, (bytes, bytearray)) else int(x).to_bytes(32, "little")


class COracle:
    def __init__(self):
        if not os.path.exists(LIB):
            raise ImportError("oracle C library not built: make -C oracle/c")
        self.lib = ctypes.CDLL(LIB)
        cp, u32 = ctypes.c_char_p, ctypes.c_uint32
        self.lib.oracle_prove.restype = ctypes.c_size_t
        self.lib.oracle_prove.argtypes = [ctypes.c_int, ctypes.POINTER(u32), cp, cp, cp, u32, cp, cp, u32, cp, cp, cp,
                                          ctypes.POINTER(u32), cp, u32]
        self.lib.oracle_gen_point.argtypes = [ctypes.c_int, u32, u32, cp]
        self.lib.oracle_msm.argtypes = [cp, cp, u32, cp]
        self.lib.oracle_warm_gens.argtypes = [u32]
        self.blob = open(BLOB, "rb").read()

    def prove(self, gadget, ip, sp, label, values, blindings, seed, want_wires=False, prove=True, aux=None):
        """-> dict(proof, comms, n, q, m[, wires]).  `aux`: the gadget's auxiliary table when it is not the Poseidon
        constants (MiMC round constants, rounds * 32 bytes)."""
        m = len(values) // 32
        if gadget in (VSMT_4, VSMT_2) and len(ip) == 2:
            ip = list(ip) + [1]   # S-box of the tree's Poseidon: Inverse unless the caller asks for Cube (ip[2] = 0)
        blob = self.blob if aux is None else aux
        ipa = (ctypes.c_uint32 * max(1, len(ip)))(*ip)
        proof = ctypes.create_string_buffer(1 + 32 * (13 + 64)) if prove else None
        comms = ctypes.create_string_buffer(32 * max(1, m))
        stats = (ctypes.c_uint32 * 3)()
        wires, cap = None, 0
        if want_wires:
            st = (ctypes.c_uint32 * 3)()
            self.lib.oracle_prove(gadget, ipa, _sc(sp), blob, label, len(label), values, blindings, m, seed, None, None, st, None, 0)
            cap = st[0]
            wires = ctypes.create_string_buffer(96 * cap)
        n = self.lib.oracle_prove(gadget, ipa, _sc(sp), blob, label, len(label), values, blindings, m, seed, proof, comms, stats, wires, cap)
        out = dict(proof=proof.raw[:n] if prove else None, comms=[comms.raw[32 * i:32 * i + 32] for i in range(m)],
                   n=stats[0], q=stats[1], m=stats[2])
        if want_wires:
            out["wires"] = wires.raw
        return out

    # bench.py helpers -----------------------------------------------------------------------
    def compile_vsmt4(self, levels, partial_rounds, root):
        N = 1 << ((583 * levels) - 1).bit_length()
        self.lib.oracle_warm_gens(N)  # generator setup is outside the timed region (reference :386-387)
        return (levels, partial_rounds, root)

    def prove_vsmt4(self, circ, values, blindings, seed):
        levels, pr, root = circ
        return self.prove(VSMT_4, [levels, pr], root, b"VSMT", values, blindings, seed)["proof"]

    def prove_case(self, gname, ip, sp, label, values, blindings, seed, **kw):
        """Front-end style call (gadget name, iparams, sparams as in include/bpr1cs_gadgets.h): MiMC gadgets carry their
        round constants as the leading sparams and the image as the last one."""
        gid = GADGET_IDS[gname]
        if gid in (MIMC, MIMC_SET_MEMBERSHIP):
            rounds = ip[0]
            aux = b"".join(_sc(x) for x in sp[:rounds])
            return self.prove(gid, ip, sp[rounds], label, values, blindings, seed, aux=aux, **kw)
        return self.prove(gid, ip, sp[0] if sp else bytes(32), label, values, blindings, seed, **kw)

    def gen_point(self, which, i, cap):
        out = ctypes.create_string_buffer(32)
        self.lib.oracle_gen_point(which, i, cap, out)
        return out.raw

    def msm(self, scalars, points):
        out = ctypes.create_string_buffer(32)
        self.lib.oracle_msm(b"".join(_sc(s) for s in scalars), b"".join(points), len(scalars), out)
        return out.raw
