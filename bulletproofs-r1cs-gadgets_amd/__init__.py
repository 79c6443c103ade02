"""bulletproofs-r1cs-gadgets_amd — MI355X-native Bulletproofs R1CS prover hot path.

Thin ctypes binding over the C ABI of include/bpr1cs.h (libbpr1cs_hip.so, built
in-tree by `__graft_entry__.build()` / `csrc/build.sh`).  The library is HIP
only: importing works without a GPU, every compute call fails loudly without
one.  Nothing here imports or calls the test oracle.

Import with importlib (the directory name carries hyphens):
    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbpr1cs_hip.so")
GADGETS_LIB_PATH = os.path.join(_HERE, "csrc", "libbpr1cs_gadgets.so")
POSEIDON_PARAMS_PATH = os.path.join(_HERE, "data", "poseidon_params_ristretto.bin")

VAR_COMMITTED, VAR_MUL_LEFT, VAR_MUL_RIGHT, VAR_MUL_OUT, VAR_ONE = 0, 1, 2, 3, 4
W_LC, W_INV_LEFT, W_BIT, W_NOTBIT = 0, 1, 2, 3

ERRORS = {0: "OK", -1: "InvalidGeneratorsLength", -2: "FormatError", -3: "VerificationError",
          -4: "MissingAssignment", -5: "GadgetError", -16: "NoDevice", -17: "InvalidArgument", -18: "DeviceError", -19: "OutOfMemory"}


class R1CSError(RuntimeError):
    def __init__(self, code):
        super().__init__("bpr1cs: %s (%d)" % (ERRORS.get(code, "?"), code))
        self.code = code


class _WOp(ctypes.Structure):
    _fields_ = [("lkind", ctypes.c_uint32), ("larg", ctypes.c_uint32), ("rkind", ctypes.c_uint32), ("rarg", ctypes.c_uint32)]


class _CircuitDesc(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint32), ("q", ctypes.c_uint32), ("m", ctypes.c_uint32),
                ("row_off", ctypes.POINTER(ctypes.c_uint32)), ("term_var", ctypes.POINTER(ctypes.c_uint32)),
                ("term_coeff", ctypes.c_char_p),
                ("wops", ctypes.POINTER(_WOp)), ("n_lc", ctypes.c_uint32),
                ("lc_off", ctypes.POINTER(ctypes.c_uint32)), ("lc_var", ctypes.POINTER(ctypes.c_uint32)),
                ("lc_coeff", ctypes.c_char_p),
                # optional Poseidon annotations (include/bpr1cs.h); left zero by this wrapper
                ("n_poseidon_params", ctypes.c_uint32), ("poseidon_params", ctypes.c_void_p),
                ("n_poseidon_perms", ctypes.c_uint32), ("poseidon_perms", ctypes.c_void_p)]


_lib = None


_B32 = ctypes.c_uint8 * 32


class ProofStruct(ctypes.Structure):
    """bpr1cs_proof (include/bpr1cs.h): the typed R1CSProof."""
    _fields_ = [(k, _B32) for k in ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6", "t_x", "t_x_blinding", "e_blinding")] + \
               [("lg_n", ctypes.c_uint32), ("L", _B32 * 32), ("R", _B32 * 32), ("ipp_a", _B32), ("ipp_b", _B32)]


# options of a generator handle (include/bpr1cs.h BPR1CS_OPT_*)
OPT_UNFOLD_ROUNDS, OPT_WITNESS_TEAM, OPT_TAIL_ROUNDS, OPT_SHARED_BACK, OPT_FACTOR_VECTORS = 0, 2, 3, 4, 5
OPT_MSM_THREADS_LOG2, OPT_JOB_PROOFS, OPT_JOBS_IN_FLIGHT, OPT_HOST_CHAIN_PROOFS, OPT_WINDOW_BITS = 6, 7, 8, 9, 16
OPTIONS = dict(unfold=OPT_UNFOLD_ROUNDS, witness_team=OPT_WITNESS_TEAM, tail_rounds=OPT_TAIL_ROUNDS, shared_back=OPT_SHARED_BACK,
               factor_vectors=OPT_FACTOR_VECTORS, msm_threads_log2=OPT_MSM_THREADS_LOG2, job_proofs=OPT_JOB_PROOFS,
               jobs_in_flight=OPT_JOBS_IN_FLIGHT, host_chain_proofs=OPT_HOST_CHAIN_PROOFS, window_bits=OPT_WINDOW_BITS)


class ProveStats(ctypes.Structure):
    """bpr1cs_prove_stats (include/bpr1cs.h)"""
    _fields_ = [("jobs", ctypes.c_uint32), ("job_proofs", ctypes.c_uint32), ("phase_ms", ctypes.c_float * 6), ("msm_ms", ctypes.c_double),
                ("msm_launches", ctypes.c_uint64), ("msm_terms", ctypes.c_uint64), ("msm_adds", ctypes.c_uint64), ("host_chains", ctypes.c_uint64),
                ("sizing_free_bytes", ctypes.c_uint64), ("sizing_bytes_per_proof", ctypes.c_uint64), ("sizing_fixed_bytes", ctypes.c_uint64)]


def load_library(path=None):
    """Load libbpr1cs_hip.so (or an ABI-compatible test build when `path` is given)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("bpr1cs: %s is missing — run __graft_entry__.build() (hipcc, gfx950). "
                          "There is no CPU fallback." % p)
    lib = ctypes.CDLL(p)
    vp, u32, sz, cp = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_char_p
    lib.bpr1cs_device_count.restype = ctypes.c_int
    lib.bpr1cs_set_device.argtypes = [ctypes.c_int]
    lib.bpr1cs_gens_create.argtypes = [u32, ctypes.POINTER(vp)]
    lib.bpr1cs_gens_destroy.argtypes = [vp]
    lib.bpr1cs_gens_capacity.argtypes = [vp]
    lib.bpr1cs_gens_capacity.restype = u32
    lib.bpr1cs_gens_point.argtypes = [vp, ctypes.c_int, u32, cp]
    lib.bpr1cs_circuit_create.argtypes = [ctypes.POINTER(_CircuitDesc), ctypes.POINTER(vp)]
    lib.bpr1cs_circuit_destroy.argtypes = [vp]
    lib.bpr1cs_proof_len.argtypes = [vp]
    lib.bpr1cs_proof_len.restype = sz
    lib.bpr1cs_prove_batch.argtypes = [vp, vp, cp, sz, cp, cp, cp, cp, sz, cp, cp]
    lib.bpr1cs_prove_batch_begin.argtypes = [vp, vp, cp, sz, cp, cp, cp, cp, sz, ctypes.POINTER(vp)]
    lib.bpr1cs_prove_batch_end.argtypes = [vp, cp, cp]
    lib.bpr1cs_verify_batch.argtypes = [vp, vp, cp, sz, cp, cp, cp, sz, ctypes.POINTER(ctypes.c_int)]
    lib.bpr1cs_msm_fixed.argtypes = [vp, ctypes.POINTER(u32), sz, cp, sz, cp]
    lib.bpr1cs_verify_batch_combined.argtypes = [vp, vp, cp, sz, cp, cp, cp, cp, ctypes.c_uint64, sz, cp, ctypes.POINTER(ctypes.c_int)]
    lib.bpr1cs_points_sum.argtypes = [cp, sz, cp]
    lib.bpr1cs_msm.argtypes = [cp, cp, sz, cp]
    lib.bpr1cs_transcript_new.argtypes = [cp, sz]
    lib.bpr1cs_transcript_new.restype = vp
    lib.bpr1cs_transcript_free.argtypes = [vp]
    lib.bpr1cs_transcript_append_message.argtypes = [vp, cp, sz, cp, sz]
    lib.bpr1cs_transcript_challenge_bytes.argtypes = [vp, cp, sz, cp, sz]
    lib.bpr1cs_poseidon_permutation_batch.argtypes = [vp, ctypes.c_int, cp, sz, cp]
    lib.bpr1cs_circuit_macro_perms.argtypes = [ctypes.c_void_p]
    lib.bpr1cs_circuit_macro_perms.restype = ctypes.c_int
    lib.bpr1cs_gens_create_opts.argtypes = [u32, ctypes.POINTER(ctypes.c_int32), sz, ctypes.POINTER(vp)]
    lib.bpr1cs_prove_batch_transcripts.argtypes = [vp, vp, ctypes.POINTER(vp), sz, cp, cp, cp, cp, sz, cp, cp]
    lib.bpr1cs_last_prove_stats.argtypes = [ctypes.POINTER(ProveStats)]
    lib.bpr1cs_gens_set_option.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.bpr1cs_gens_release_scratch.argtypes = [vp]
    lib.bpr1cs_gens_table_info.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_uint64)]
    lib.bpr1cs_verify_batch_scalars.argtypes = [vp, vp, cp, sz, cp, cp, cp, cp, ctypes.c_uint64, sz, cp, cp, ctypes.POINTER(ctypes.c_int)]
    lib.bpr1cs_scalars_sum.argtypes = [cp, sz, sz, cp]
    lib.bpr1cs_comm_unique_id.argtypes = [cp]
    lib.bpr1cs_comm_create.argtypes = [cp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    lib.bpr1cs_comm_wrap.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    lib.bpr1cs_comm_destroy.argtypes = [vp]
    lib.bpr1cs_verify_batch_sharded.argtypes = [vp, vp, cp, sz, cp, cp, cp, cp, ctypes.c_uint64, sz, vp, ctypes.POINTER(ctypes.c_int)]
    lib.bpr1cs_ipa_create.argtypes = [vp, vp, cp, cp, cp, cp, cp, sz, cp, cp, cp, cp]
    lib.bpr1cs_proof_parse.argtypes = [cp, sz, ctypes.POINTER(ProofStruct)]
    lib.bpr1cs_proof_serialize.argtypes = [ctypes.POINTER(ProofStruct), cp, sz, ctypes.POINTER(sz)]
    lib.bpr1cs_proof_serialized_len.argtypes = [ctypes.POINTER(ProofStruct)]
    lib.bpr1cs_proof_serialized_len.restype = sz
    lib.bpr1cs_device_rates.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    if path is None:
        _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise R1CSError(rc)


def _u32arr(xs):
    return (ctypes.c_uint32 * max(1, len(xs)))(*xs)


class Gens:
    """PedersenGens::default() + BulletproofGens::new(capacity, 1)."""

    def __init__(self, capacity, lib=None, **options):
        """options: names of OPTIONS (unfold=4, window_bits=8, job_proofs=1024 ...); defaults = the benchmarked configuration"""
        self.lib = lib or load_library()
        h = ctypes.c_void_p()
        pairs = []
        for k, v in options.items():
            pairs += [OPTIONS[k], int(v)]
        arr = (ctypes.c_int32 * max(1, len(pairs)))(*pairs)
        _chk(self.lib.bpr1cs_gens_create_opts(capacity, arr, len(pairs) // 2, ctypes.byref(h)))
        self.h, self.capacity = h, capacity

    def point(self, which, i=0):
        out = ctypes.create_string_buffer(32)
        _chk(self.lib.bpr1cs_gens_point(self.h, which, i, out))
        return out.raw

    def msm_fixed(self, bases, scalars, batch):
        """scalars: bytes batch*terms*32 (proof-major) -> batch compressed points."""
        out = ctypes.create_string_buffer(32 * batch)
        _chk(self.lib.bpr1cs_msm_fixed(self.h, _u32arr(bases), len(bases), scalars, batch, out))
        raw = out.raw
        return [raw[32 * i:32 * i + 32] for i in range(batch)]

    def set_option(self, option, value):
        """option: an OPT_* constant or a name of OPTIONS; value < 0: back to the default"""
        _chk(self.lib.bpr1cs_gens_set_option(self.h, OPTIONS.get(option, option) if isinstance(option, str) else option, value))

    def release_scratch(self):
        """drop the handle's back-phase arena and the allocator's cache (before a job of a very different shape)"""
        _chk(self.lib.bpr1cs_gens_release_scratch(self.h))
        self.lib.bpr1cs_release_cached_memory()

    def table_info(self):
        """-> dict(window_bits, windows, format, bytes) of the fixed-base tables"""
        w, k, f, b = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        _chk(self.lib.bpr1cs_gens_table_info(self.h, ctypes.byref(w), ctypes.byref(k), ctypes.byref(f), ctypes.byref(b)))
        return dict(window_bits=w.value, windows=k.value, format=f.value, bytes=b.value)

    def close(self):
        if self.h:
            self.lib.bpr1cs_gens_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Circuit:
    """Flattened constraint system (+ optional device witness program).

    constraints: list of rows, each a list of ((kind, index), coeff_int_or_bytes)
    wops: optional list of (lkind, larg, rkind, rarg); lcs: list of term lists for W_LC operands.
    """

    def __init__(self, n, m, constraints, wops=None, lcs=None, lib=None, raw=None):
        self.lib = lib or load_library()
        self.n, self.m = n, m
        if raw is not None:
            row_off, tvar, tcoef, q = raw
        else:
            row_off, tvar, tcoef = [0], [], bytearray()
            for row in constraints:
                for (kind, idx), c in row:
                    tvar.append((kind << 28) | idx)
                    tcoef += c if isinstance(c, (bytes, bytearray)) else int(c).to_bytes(32, "little")
                row_off.append(len(tvar))
            q = len(constraints)
        self.q = q
        d = _CircuitDesc()
        d.n, d.q, d.m = n, q, m
        self._keep = [_u32arr(row_off), _u32arr(tvar), bytes(tcoef) or b"\0"]
        d.row_off, d.term_var, d.term_coeff = self._keep[0], self._keep[1], self._keep[2]
        if wops is not None:
            arr = (_WOp * max(1, len(wops)))(*[_WOp(*w) for w in wops])
            lc_off, lc_var, lc_coeff = [0], [], bytearray()
            for terms in (lcs or []):
                for (kind, idx), c in terms:
                    lc_var.append((kind << 28) | idx)
                    lc_coeff += c if isinstance(c, (bytes, bytearray)) else int(c).to_bytes(32, "little")
                lc_off.append(len(lc_var))
            self._keep += [arr, _u32arr(lc_off), _u32arr(lc_var), bytes(lc_coeff) or b"\0"]
            d.wops, d.n_lc = arr, len(lcs or [])
            d.lc_off, d.lc_var, d.lc_coeff = self._keep[4], self._keep[5], self._keep[6]
        h = ctypes.c_void_p()
        _chk(self.lib.bpr1cs_circuit_create(ctypes.byref(d), ctypes.byref(h)))
        self.h = h
        self.proof_len = self.lib.bpr1cs_proof_len(h)

    def close(self):
        if self.h:
            self.lib.bpr1cs_circuit_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_batch(gens, circuit, label, values, v_blindings, rng_seeds, batch, wires=None):
    """-> (list of proof bytes, list of per-proof commitment lists).

    values / v_blindings: bytes batch*m*32 proof-major; rng_seeds: bytes batch*32;
    wires: None (device witness program) or bytes batch*3*n*32 (a_L|a_R|a_O per proof).
    """
    lib = gens.lib
    m, plen = circuit.m, circuit.proof_len
    assert len(values) == batch * m * 32 and len(v_blindings) == batch * m * 32 and len(rng_seeds) == batch * 32
    if wires is not None:
        assert len(wires) == batch * 3 * circuit.n * 32
    proofs = ctypes.create_string_buffer(batch * plen)
    comms = ctypes.create_string_buffer(max(1, batch * m * 32))
    _chk(lib.bpr1cs_prove_batch(gens.h, circuit.h, label, len(label), values or b"\0", v_blindings or b"\0", rng_seeds,
                                wires, batch, proofs, comms))
    praw, craw = proofs.raw, comms.raw  # .raw copies the whole buffer: take it once
    P = [praw[i * plen:(i + 1) * plen] for i in range(batch)]
    C = [[craw[(i * m + j) * 32:(i * m + j + 1) * 32] for j in range(m)] for i in range(batch)]
    return P, C


def verify_batch(gens, circuit, label, proofs, commitments, batch, seeds=None):
    """proofs: list of bytes or concatenated bytes; commitments: list of lists (each m x 32 bytes) or bytes.
    -> list of bool (Verifier::verify accepted).  seeds: batch*32 bytes for the verifier's TranscriptRng
    (default: fresh os.urandom, as upstream's thread_rng)."""
    if seeds is None:
        seeds = os.urandom(32 * batch)
    pf = proofs if isinstance(proofs, (bytes, bytearray)) else b"".join(proofs)
    cm = commitments if isinstance(commitments, (bytes, bytearray)) else b"".join(b"".join(c) for c in commitments)
    assert len(pf) == batch * circuit.proof_len and len(cm) == batch * circuit.m * 32
    ok = (ctypes.c_int * batch)()
    _chk(gens.lib.bpr1cs_verify_batch(gens.h, circuit.h, label, len(label), pf, cm or b"\0", seeds, batch, ok))
    return [bool(x) for x in ok]


def verify_batch_combined(gens, circuit, label, proofs, commitments, batch, batch_seed=None, index_base=0, seeds=None):
    """Cross-proof batched mega-check of this caller's `batch` proofs -> (partial point: 32 bytes, wellformed: bool).
    The whole job is accepted iff points_sum_is_identity(all callers' points) and all callers were well-formed.
    batch_seed / seeds default to fresh os.urandom (the weights must not be predictable, see include/bpr1cs.h)."""
    if batch_seed is None:
        batch_seed = os.urandom(32)
    if seeds is None:
        seeds = os.urandom(32 * batch)
    pf = proofs if isinstance(proofs, (bytes, bytearray)) else b"".join(proofs)
    cm = commitments if isinstance(commitments, (bytes, bytearray)) else b"".join(b"".join(c) for c in commitments)
    assert len(pf) == batch * circuit.proof_len and len(cm) == batch * circuit.m * 32 and len(batch_seed) == 32
    out = ctypes.create_string_buffer(32)
    wf = ctypes.c_int()
    _chk(gens.lib.bpr1cs_verify_batch_combined(gens.h, circuit.h, label, len(label), pf, cm or b"\0", seeds, batch_seed, index_base, batch,
                                               out, ctypes.byref(wf)))
    return out.raw, bool(wf.value)


def verify_batch_scalars(gens, circuit, label, proofs, commitments, batch, batch_seed=None, index_base=0, seeds=None):
    """First half of the multi-GPU batched verifier -> (combined scalar vector: (2N+2)*32 bytes in base order
    B, B~, G.., H.., own-points sum: 32 bytes, wellformed)."""
    if batch_seed is None:
        batch_seed = os.urandom(32)
    if seeds is None:
        seeds = os.urandom(32 * batch)
    pf = proofs if isinstance(proofs, (bytes, bytearray)) else b"".join(proofs)
    cm = commitments if isinstance(commitments, (bytes, bytearray)) else b"".join(b"".join(c) for c in commitments)
    N = 1 << max(0, (circuit.n - 1).bit_length())
    out = ctypes.create_string_buffer(32 * (2 * N + 2))
    own = ctypes.create_string_buffer(32)
    wf = ctypes.c_int()
    _chk(gens.lib.bpr1cs_verify_batch_scalars(gens.h, circuit.h, label, len(label), pf, cm or b"\0", seeds, batch_seed, index_base, batch,
                                              out, own, ctypes.byref(wf)))
    return out.raw, own.raw, bool(wf.value)


class Comm:
    """An RCCL communicator owned by the library (include/bpr1cs.h: bpr1cs_comm).  `unique_id()` on rank 0, hand the 128
    bytes to the other ranks (any side channel, e.g. a torch.distributed broadcast), `Comm(id, rank, world)` on every rank."""

    def __init__(self, uid, rank, world, lib=None):
        self.lib = lib or load_library()
        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        _chk(self.lib.bpr1cs_comm_create(uid, rank, world, ctypes.byref(h)))
        self.h = h

    @staticmethod
    def unique_id(lib=None):
        lib = lib or load_library()
        buf = ctypes.create_string_buffer(128)
        _chk(lib.bpr1cs_comm_unique_id(buf))
        return buf.raw

    def close(self):
        if self.h:
            self.lib.bpr1cs_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def verify_batch_sharded(gens, circuit, label, proofs, commitments, batch, comm=None, batch_seed=None, index_base=0, seeds=None):
    """The multi-GPU batched verifier in one call per rank (bpr1cs_verify_batch_sharded: RCCL all_gathers inside the library)
    -> True / False, the same on every rank.  comm = None: a job of one rank."""
    if batch_seed is None:
        batch_seed = os.urandom(32)
    if seeds is None:
        seeds = os.urandom(32 * batch)
    pf = proofs if isinstance(proofs, (bytes, bytearray)) else b"".join(proofs)
    cm = commitments if isinstance(commitments, (bytes, bytearray)) else b"".join(b"".join(c) for c in commitments)
    ok = ctypes.c_int()
    _chk(gens.lib.bpr1cs_verify_batch_sharded(gens.h, circuit.h, label, len(label), pf, cm or b"\0", seeds, batch_seed, index_base, batch,
                                              comm.h if comm is not None else None, ctypes.byref(ok)))
    return bool(ok.value)


def scalars_sum(vectors, lib=None):
    """element-wise sum mod l of equally long vectors of canonical scalars (bytes each) -> bytes"""
    lib = lib or load_library()
    n = len(vectors[0]) // 32
    out = ctypes.create_string_buffer(32 * n)
    _chk(lib.bpr1cs_scalars_sum(b"".join(vectors), len(vectors), n, out))
    return out.raw


def ipa_create(gens, transcript, Q, G_factors, H_factors, a, b):
    """InnerProductProof::create on the device over gens.G/H[0..n) -> (L list, R list, a, b) ; `transcript` (Transcript) advances."""
    n = len(a)
    lg = max(0, n.bit_length() - 1)
    Lb, Rb = ctypes.create_string_buffer(32 * max(1, lg)), ctypes.create_string_buffer(32 * max(1, lg))
    ao, bo = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    j = lambda xs: b"".join(_sc(x) for x in xs)
    _chk(gens.lib.bpr1cs_ipa_create(gens.h, transcript.h, Q, j(G_factors), j(H_factors), j(a), j(b), n, Lb, Rb, ao, bo))
    return ([Lb.raw[32 * i:32 * i + 32] for i in range(lg)], [Rb.raw[32 * i:32 * i + 32] for i in range(lg)],
            int.from_bytes(ao.raw, "little"), int.from_bytes(bo.raw, "little"))


_PROOF_FIELDS = ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6", "t_x", "t_x_blinding", "e_blinding", "ipp_a", "ipp_b")


def proof_parse(data, lib=None):
    """R1CSProof::from_bytes -> dict of 32-byte fields + lists L, R (raises R1CSError(FormatError))."""
    lib = lib or load_library()
    ps = ProofStruct()
    _chk(lib.bpr1cs_proof_parse(data, len(data), ctypes.byref(ps)))
    d = {k: bytes(getattr(ps, k)) for k in _PROOF_FIELDS}
    d["L"] = [bytes(ps.L[i]) for i in range(ps.lg_n)]
    d["R"] = [bytes(ps.R[i]) for i in range(ps.lg_n)]
    return d


def proof_serialize(d, lib=None):
    """R1CSProof::to_bytes of a dict as returned by proof_parse."""
    lib = lib or load_library()
    ps = ProofStruct()
    for k in _PROOF_FIELDS:
        setattr(ps, k, _B32(*d[k]))
    ps.lg_n = len(d["L"])
    for i, (l, r) in enumerate(zip(d["L"], d["R"])):
        ps.L[i] = _B32(*l)
        ps.R[i] = _B32(*r)
    out = ctypes.create_string_buffer(lib.bpr1cs_proof_serialized_len(ctypes.byref(ps)))
    n = ctypes.c_size_t()
    _chk(lib.bpr1cs_proof_serialize(ctypes.byref(ps), out, len(out), ctypes.byref(n)))
    return out.raw[:n.value]


def device_rates(seconds_each=0.08, lib=None):
    """-> (v_mad_i64_i32 lane-ops/s, ge_madd_t table adds/s), sustained, measured on the current device"""
    lib = lib or load_library()
    a, b = ctypes.c_double(), ctypes.c_double()
    _chk(lib.bpr1cs_device_rates(seconds_each, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def release_cached_memory(lib=None):
    (lib or load_library()).bpr1cs_release_cached_memory()


class _PoseidonParams(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("full_rounds_beginning", ctypes.c_uint32), ("partial_rounds", ctypes.c_uint32),
                ("full_rounds_end", ctypes.c_uint32), ("mds", ctypes.c_char_p), ("round_keys", ctypes.c_char_p)]


def poseidon_permutation_batch(states, inverse=True, partial_rounds=140, lib=None):
    """Device bulk form of the reference's Poseidon_permutation (width 6, 4 + partial_rounds + 4 rounds, the
    constants of poseidon_constants.rs): states = list of 6-lists of ints -> list of 6-lists of ints."""
    lib = lib or load_library()
    blob = poseidon_blob()
    nk = (8 + partial_rounds) * 6
    pp = _PoseidonParams(6, 4, partial_rounds, 4, blob[:36 * 32], blob[36 * 32:36 * 32 + nk * 32])
    n = len(states)
    inp = b"".join(_sc(x) for st in states for x in st)
    out = ctypes.create_string_buffer(n * 6 * 32)
    _chk(lib.bpr1cs_poseidon_permutation_batch(ctypes.byref(pp), 1 if inverse else 0, inp, n, out))
    return [[int.from_bytes(out.raw[32 * (6 * h + i):32 * (6 * h + i) + 32], "little") for i in range(6)] for h in range(n)]


def msm(scalars, points, lib=None):
    """sum_i scalars[i] * points[i] over arbitrary compressed ristretto points (ints / 32-byte strings) -> 32 bytes"""
    lib = lib or load_library()
    out = ctypes.create_string_buffer(32)
    _chk(lib.bpr1cs_msm(b"".join(_sc(x) for x in scalars), b"".join(points), len(points), out))
    return out.raw


class Transcript:
    """merlin::Transcript through the C ABI (host side)."""

    def __init__(self, label, lib=None):
        self.lib = lib or load_library()
        self.h = self.lib.bpr1cs_transcript_new(label, len(label))

    def append_message(self, label, msg):
        self.lib.bpr1cs_transcript_append_message(self.h, label, len(label), msg, len(msg))

    def challenge_bytes(self, label, n):
        out = ctypes.create_string_buffer(n)
        self.lib.bpr1cs_transcript_challenge_bytes(self.h, label, len(label), out, n)
        return out.raw

    def __del__(self):
        try:
            self.lib.bpr1cs_transcript_free(self.h)
        except Exception:
            pass


def points_sum(points, lib=None):
    """compressed ristretto points -> compressed sum (raises FormatError if one does not decode)"""
    lib = lib or load_library()
    out = ctypes.create_string_buffer(32)
    _chk(lib.bpr1cs_points_sum(b"".join(points), len(points), out))
    return out.raw


def points_sum_is_identity(points, lib=None):
    return points_sum(points, lib=lib) == bytes(32)


class ProveJob:
    """An in-flight bpr1cs_prove_batch_begin; .finish() -> (proofs, commitments)."""

    def __init__(self, gens, circuit, label, values, v_blindings, rng_seeds, batch, wires=None):
        self.lib, self.batch, self.m, self.plen = gens.lib, batch, circuit.m, circuit.proof_len
        h = ctypes.c_void_p()
        _chk(self.lib.bpr1cs_prove_batch_begin(gens.h, circuit.h, label, len(label), values or b"\0", v_blindings or b"\0", rng_seeds,
                                               wires, batch, ctypes.byref(h)))
        self.h = h

    def finish(self, split=True):
        proofs = ctypes.create_string_buffer(self.batch * self.plen)
        comms = ctypes.create_string_buffer(max(1, self.batch * self.m * 32))
        h, self.h = self.h, None
        _chk(self.lib.bpr1cs_prove_batch_end(h, proofs, comms))
        praw, craw = proofs.raw, comms.raw
        if not split:
            return praw, craw
        m, plen = self.m, self.plen
        return ([praw[i * plen:(i + 1) * plen] for i in range(self.batch)],
                [[craw[(i * m + j) * 32:(i * m + j + 1) * 32] for j in range(m)] for i in range(self.batch)])


def last_prove_stats(lib=None):
    """bpr1cs_prove_stats of the last prove call that returned on this thread -> dict"""
    lib = lib or load_library()
    st = ProveStats()
    _chk(lib.bpr1cs_last_prove_stats(ctypes.byref(st)))
    return dict(jobs=st.jobs, job_proofs=st.job_proofs, phase_ms=list(st.phase_ms), msm_ms=st.msm_ms, msm_launches=st.msm_launches, msm_terms=st.msm_terms, msm_adds=st.msm_adds, host_chains=st.host_chains,
                sizing_free_bytes=st.sizing_free_bytes, sizing_bytes_per_proof=st.sizing_bytes_per_proof, sizing_fixed_bytes=st.sizing_fixed_bytes)


def prove_batch_transcripts(gens, circuit, transcripts, values, v_blindings, rng_seeds, batch, wires=None):
    """bpr1cs_prove_batch_transcripts: `transcripts` = one Transcript (every proof starts from a copy) or `batch` of them (each is
    advanced to the state upstream's `&mut transcript` has after prove()) -> (proofs, commitments) as prove_batch"""
    lib = gens.lib
    ts = transcripts if isinstance(transcripts, (list, tuple)) else [transcripts]
    arr = (ctypes.c_void_p * len(ts))(*[t.h for t in ts])
    m, plen = circuit.m, circuit.proof_len
    proofs = ctypes.create_string_buffer(batch * plen)
    comms = ctypes.create_string_buffer(max(1, batch * m * 32))
    _chk(lib.bpr1cs_prove_batch_transcripts(gens.h, circuit.h, arr, len(ts), values or b"\0", v_blindings or b"\0", rng_seeds, wires, batch, proofs, comms))
    praw, craw = proofs.raw, comms.raw
    return ([praw[i * plen:(i + 1) * plen] for i in range(batch)],
            [[craw[(i * m + j) * 32:(i * m + j + 1) * 32] for j in range(m)] for i in range(batch)])


def output_buffers(circuit, batch):
    """caller-owned output memory for prove_batch_raw(out=...): (proofs, commitments) as writable bytearrays, pages touched -
    what a C caller hands to bpr1cs_prove_batch (a timed region then holds the library call, not the allocation of its result)"""
    bufs = bytearray(batch * circuit.proof_len), bytearray(max(1, batch * circuit.m * 32))
    for b in bufs:   # first touch of every page now, not inside the caller's timed region
        b[0::4096] = bytes(len(range(0, len(b), 4096)))
    return bufs


def prove_batch_raw(gens, circuit, label, values, v_blindings, rng_seeds, batch, out=None):
    """ONE bpr1cs_prove_batch call over any batch (the library cuts it into device jobs) -> (proof bytes, commitment bytes), unsplit.
    out: (proofs, commitments) from output_buffers() - filled in place and returned as they are (no copy)"""
    m, plen = circuit.m, circuit.proof_len
    if out is not None:
        pbuf, cbuf = out
        assert len(pbuf) >= batch * plen and len(cbuf) >= max(1, batch * m * 32)
        proofs = (ctypes.c_char * len(pbuf)).from_buffer(pbuf)
        comms = (ctypes.c_char * len(cbuf)).from_buffer(cbuf)
        _chk(gens.lib.bpr1cs_prove_batch(gens.h, circuit.h, label, len(label), values or b"\0", v_blindings or b"\0", rng_seeds, None, batch, proofs, comms))
        del proofs, comms
        return pbuf, cbuf
    proofs = ctypes.create_string_buffer(batch * plen)
    comms = ctypes.create_string_buffer(max(1, batch * m * 32))
    _chk(gens.lib.bpr1cs_prove_batch(gens.h, circuit.h, label, len(label), values or b"\0", v_blindings or b"\0", rng_seeds, None, batch, proofs, comms))
    return proofs.raw, comms.raw


# ---------------------------------------------------------------- host front-end (libbpr1cs_gadgets.so)
_glib = None


def poseidon_blob():
    with open(POSEIDON_PARAMS_PATH, "rb") as f:
        return f.read()


def load_gadgets_library(path=None):
    """C++ mirror of the reference's gadget layer (include/bpr1cs_gadgets.h)."""
    global _glib
    if _glib is not None and path is None:
        return _glib
    p = path or GADGETS_LIB_PATH
    if not os.path.exists(p):
        raise ImportError("bpr1cs: %s is missing — run __graft_entry__.build()" % p)
    if path is None:
        load_library()  # resolve libbpr1cs_hip.so first
    g = ctypes.CDLL(p)
    vp, u32, sz, cp, ip = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32)
    g.bpr1cs_gadget_compile.argtypes = [cp, ip, sz, cp, sz, cp, sz, ctypes.POINTER(vp), ip, ip, ip, ctypes.POINTER(ctypes.c_int)]
    g.bpr1cs_gadget_prove_single.argtypes = [cp, ip, sz, cp, sz, cp, sz, u32, cp, sz, cp, cp, sz, cp, cp, sz, ctypes.POINTER(sz), cp]
    g.bpr1cs_gadget_prove_on.argtypes = [vp, cp, ip, sz, cp, sz, cp, sz, cp, sz, cp, cp, sz, sz, cp, cp, sz, ctypes.POINTER(sz), cp, ctypes.POINTER(ctypes.c_double)]
    g.bpr1cs_gadget_prove_on_flags.argtypes = g.bpr1cs_gadget_prove_on.argtypes + [ctypes.c_uint32]
    g.bpr1cs_gadget_verify_on.argtypes = [vp, cp, ip, sz, cp, sz, cp, sz, cp, sz, cp, sz, cp, sz, ctypes.POINTER(ctypes.c_double)]
    g.bpr1cs_gadget_synthesize.argtypes = [cp, ip, sz, cp, sz, cp, sz, cp, sz, cp, sz, ip, ip]
    g.bpr1cs_gadget_verify_single.argtypes = [cp, ip, sz, cp, sz, cp, sz, u32, cp, sz, cp, sz, cp, sz]
    g.bpr1cs_poseidon_hash.argtypes = [ctypes.c_int, ctypes.c_int, u32, cp, sz, cp, cp]
    g.bpr1cs_mimc.argtypes = [cp, cp, cp, sz, cp]
    for nm in ("vsmt4", "vsmt2"):
        getattr(g, "bpr1cs_%s_new" % nm).argtypes = [u32, u32, cp, sz, ctypes.POINTER(vp)]
        getattr(g, "bpr1cs_%s_new_sbox" % nm).argtypes = [u32, u32, ctypes.c_int, cp, sz, ctypes.POINTER(vp)]
        getattr(g, "bpr1cs_%s_free" % nm).argtypes = [vp]
        getattr(g, "bpr1cs_%s_root" % nm).argtypes = [vp, cp]
        getattr(g, "bpr1cs_%s_update" % nm).argtypes = [vp, cp, cp]
        getattr(g, "bpr1cs_%s_get" % nm).argtypes = [vp, cp, cp, cp]
    for nm in ("vsmt4", "vsmt2"):
        getattr(g, "bpr1cs_%s_update_many" % nm).argtypes = [vp, cp, cp, sz]
        getattr(g, "bpr1cs_%s_get_many" % nm).argtypes = [vp, cp, sz, cp, cp]
    if path is None:
        _glib = g
    return g


def _sc(x):
    return x if isinstance(x, (bytes, bytearray)) else int(x).to_bytes(32, "little")


class CompiledGadget:
    """A reference gadget compiled (shape only) into a device circuit + witness program."""

    def __init__(self, name, iparams=(), sparams=(), lib=None, glib=None):
        self.lib = lib or load_library()
        self.glib = glib or load_gadgets_library()
        blob = poseidon_blob()
        sp = b"".join(_sc(s) for s in sparams)
        h = ctypes.c_void_p()
        n, q, m, hw = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
        _chk(self.glib.bpr1cs_gadget_compile(name.encode(), _u32arr(list(iparams)), len(iparams), sp or b"\0", len(sparams), blob, len(blob),
                                             ctypes.byref(h), ctypes.byref(n), ctypes.byref(q), ctypes.byref(m), ctypes.byref(hw)))
        self.h, self.n, self.q, self.m, self.has_witness_program = h, n.value, q.value, m.value, bool(hw.value)
        self.proof_len = self.lib.bpr1cs_proof_len(h)

    def close(self):
        if self.h:
            self.lib.bpr1cs_circuit_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_single(name, iparams, sparams, gens_capacity, label, values, blindings, rng_seed, glib=None):
    """The reference's single-proof harness over the C++ `Prover` (host synthesis, device prove)."""
    g = glib or load_gadgets_library()
    blob = poseidon_blob()
    sp = b"".join(_sc(s) for s in sparams)
    m = len(values)
    proof = ctypes.create_string_buffer(1 + 32 * (13 + 2 * 32))
    plen = ctypes.c_size_t()
    comms = ctypes.create_string_buffer(32 * max(1, m))
    _chk(g.bpr1cs_gadget_prove_single(name.encode(), _u32arr(list(iparams)), len(iparams), sp or b"\0", len(sparams), blob, len(blob),
                                      gens_capacity, label, len(label), b"".join(_sc(v) for v in values), b"".join(_sc(v) for v in blindings),
                                      m, rng_seed, proof, len(proof), ctypes.byref(plen), comms))
    return proof.raw[:plen.value], [comms.raw[32 * i:32 * i + 32] for i in range(m)]


GADGET_EAGER_COMMITS = 1
GADGET_NO_CHAIN_AHEAD = 2


def gadget_prove_on(gens, name, iparams, sparams, label, values, blindings, m, batch, rng_seeds, glib=None, eager_commits=False, chain_ahead=True):
    """bpr1cs_gadget_prove_on[_flags]: the reference's call shape (Prover::new -> commit x m -> gadget on the host -> prove) on generators
    created once; batch > 1 = one host synthesis per witness, ONE device prove with host wires.  eager_commits: every commit() computes
    its point at once, one device call each (upstream's signature; default: resolved when read).  chain_ahead=False: the C++ Prover does
    not run the proof's TranscriptRng chain beside the synthesis (the library then hashes it inside the prove call).
    -> (proofs, commitments per proof, dict of seconds: commit / gadget / circuit / prove / total)"""
    g = glib or load_gadgets_library()
    blob = poseidon_blob()
    sp = b"".join(_sc(s) for s in sparams)
    cap = 1 + 32 * (13 + 2 * 32)
    proofs = ctypes.create_string_buffer(cap * batch)
    plen = ctypes.c_size_t()
    comms = ctypes.create_string_buffer(32 * max(1, m) * batch)
    sec = (ctypes.c_double * 5)()
    _chk(g.bpr1cs_gadget_prove_on_flags(gens.h, name.encode(), _u32arr(list(iparams)), len(iparams), sp or b"\0", len(sparams), blob, len(blob),
                                        label, len(label), values or b"\0", blindings or b"\0", m, batch, rng_seeds, proofs, cap, ctypes.byref(plen), comms, sec,
                                        (GADGET_EAGER_COMMITS if eager_commits else 0) | (0 if chain_ahead else GADGET_NO_CHAIN_AHEAD)))
    n, praw, craw = plen.value, proofs.raw, comms.raw
    return ([praw[i * n:(i + 1) * n] for i in range(batch)],
            [[craw[(i * m + j) * 32:(i * m + j + 1) * 32] for j in range(m)] for i in range(batch)],
            dict(zip(("commit", "gadget", "circuit", "prove", "total"), sec)))


def gadget_verify_on(gens, name, iparams, sparams, label, proof, commitments, glib=None):
    """bpr1cs_gadget_verify_on: Verifier::new -> commit(V) x m -> gadget -> verify of ONE proof on generators created once
    -> (True / False, dict of seconds: gadget / verify / total)"""
    g = glib or load_gadgets_library()
    blob = poseidon_blob()
    sp = b"".join(_sc(s) for s in sparams)
    cm = b"".join(commitments)
    sec = (ctypes.c_double * 3)()
    rc = g.bpr1cs_gadget_verify_on(gens.h, name.encode(), _u32arr(list(iparams)), len(iparams), sp or b"\0", len(sparams), blob, len(blob),
                                   label, len(label), proof, len(proof), cm or b"\0", len(commitments), sec)
    if rc not in (0, -2, -3):
        raise R1CSError(rc)
    return rc == 0, dict(zip(("gadget", "verify", "total"), sec))


def gadget_synthesize(name, iparams, sparams, values, m, glib=None):
    """bpr1cs_gadget_synthesize: host synthesis alone -> (wires a_L | a_R | a_O as bytes, n, q); no device call"""
    g = glib or load_gadgets_library()
    blob = poseidon_blob()
    sp = b"".join(_sc(s) for s in sparams)
    n, q = ctypes.c_uint32(), ctypes.c_uint32()
    args = (name.encode(), _u32arr(list(iparams)), len(iparams), sp or b"\0", len(sparams), blob, len(blob), values or b"\0", m)
    _chk(g.bpr1cs_gadget_synthesize(*args, None, 0, ctypes.byref(n), ctypes.byref(q)))
    wires = ctypes.create_string_buffer(96 * max(1, n.value))
    _chk(g.bpr1cs_gadget_synthesize(*args, wires, 96 * n.value, ctypes.byref(n), ctypes.byref(q)))
    return wires.raw[:96 * n.value], n.value, q.value


def verify_single(name, iparams, sparams, gens_capacity, label, proof, commitments, glib=None):
    """The reference's verifier harness over the C++ `Verifier` -> True / False (VerificationError, FormatError)."""
    g = glib or load_gadgets_library()
    blob = poseidon_blob()
    sp = b"".join(_sc(s) for s in sparams)
    rc = g.bpr1cs_gadget_verify_single(name.encode(), _u32arr(list(iparams)), len(iparams), sp or b"\0", len(sparams), blob, len(blob),
                                       gens_capacity, label, len(label), proof, len(proof), b"".join(commitments), len(commitments))
    if rc in (0,):
        return True
    if rc in (-2, -3):
        return False
    raise R1CSError(rc)


def poseidon_hash(arity, inverse, partial_rounds, inputs, glib=None):
    g = glib or load_gadgets_library()
    blob = poseidon_blob()
    out = ctypes.create_string_buffer(192 if arity == 6 else 32)
    _chk(g.bpr1cs_poseidon_hash(arity, 1 if inverse else 0, partial_rounds, blob, len(blob), b"".join(_sc(x) for x in inputs), out))
    return out.raw


class SparseMerkleTree:
    """VanillaSparseMerkleTree_4 (arity 4) / VanillaSparseMerkleTree (arity 2) of the reference."""

    def __init__(self, arity, levels, partial_rounds=140, glib=None, inverse=True):
        """inverse=False: the tree over Poseidon with the Cube S-box (the reference hard-wires Inverse)"""
        self.g = glib or load_gadgets_library()
        self.nm = "vsmt4" if arity == 4 else "vsmt2"
        self.arity, self.levels = arity, levels
        blob = poseidon_blob()
        h = ctypes.c_void_p()
        _chk(getattr(self.g, "bpr1cs_%s_new_sbox" % self.nm)(levels, partial_rounds, 1 if inverse else 0, blob, len(blob), ctypes.byref(h)))
        self.h = h

    def root(self):
        out = ctypes.create_string_buffer(32)
        getattr(self.g, "bpr1cs_%s_root" % self.nm)(self.h, out)
        return out.raw

    def update(self, idx, val):
        getattr(self.g, "bpr1cs_%s_update" % self.nm)(self.h, _sc(idx), _sc(val))

    def update_many(self, leaves):
        """[(idx, val), ...] with DISTINCT indices: every tree level is hashed by one device launch."""
        idx = b"".join(_sc(i) for i, _ in leaves)
        vals = b"".join(_sc(v) for _, v in leaves)
        _chk(getattr(self.g, "bpr1cs_%s_update_many" % self.nm)(self.h, idx, vals, len(leaves)))

    def get_many(self, indices):
        """-> (leaves: bytes count*32, paths: bytes count*levels*(arity-1)*32); no re-hashing of the paths."""
        n = len(indices)
        per = 3 if self.arity == 4 else 1
        leaves = ctypes.create_string_buffer(32 * n)
        proofs = ctypes.create_string_buffer(32 * per * self.levels * n)
        _chk(getattr(self.g, "bpr1cs_%s_get_many" % self.nm)(self.h, b"".join(_sc(i) for i in indices), n, leaves, proofs))
        return leaves.raw, proofs.raw

    def get(self, idx):
        """-> (leaf bytes, [node bytes...]) with the Merkle path root level first."""
        per = 3 if self.arity == 4 else 1
        leaf = ctypes.create_string_buffer(32)
        proof = ctypes.create_string_buffer(32 * per * self.levels)
        _chk(getattr(self.g, "bpr1cs_%s_get" % self.nm)(self.h, _sc(idx), leaf, proof))
        return leaf.raw, [proof.raw[32 * i:32 * i + 32] for i in range(per * self.levels)]

    def __del__(self):
        try:
            getattr(self.g, "bpr1cs_%s_free" % self.nm)(self.h)
        except Exception:
            pass
