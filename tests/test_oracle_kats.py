"""Pin the oracle (oracle/pyref) to every published known-answer available for this path, and to
the reference's own test assertions (prove -> verify acceptance; circuit shapes)."""
import hashlib
import json
import os

import pytest

from pyref.ed import BASEPOINT, L, Point, decompress, from_uniform_bytes
from pyref.merlin import Transcript, VerificationError, sha3_512, shake256
from pyref import scenarios as S, gadgets as g
from pyref.r1cs import R1CSError
import common

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proofs.json")))


def test_rfc9496_basepoint_multiples():
    exp = ["0000000000000000000000000000000000000000000000000000000000000000",
           "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
           "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
           "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
           "da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57"]
    for k, e in enumerate(exp):
        assert (BASEPOINT * k).compress().hex() == e
        assert decompress(bytes.fromhex(e)) == BASEPOINT * k


def test_rfc9496_hash_to_group():
    h = hashlib.sha512(b"Ristretto is traditionally a short shot of espresso coffee").digest()
    assert from_uniform_bytes(h).compress().hex() == "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"


def test_rfc9496_bad_encodings_rejected():
    for bad in ["00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",  # non-canonical
                "0100000000000000000000000000000000000000000000000000000000000000",  # negative
                "edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f"]:
        assert decompress(bytes.fromhex(bad)) is None


def test_fips202_and_merlin_kat():
    assert sha3_512(b"abc") == hashlib.sha3_512(b"abc").digest()
    assert shake256(b"abc", 500) == hashlib.shake_256(b"abc").digest(500)
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_pedersen_and_bulletproof_generators():
    assert common.PC.B_blinding.compress().hex() == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"
    o = common.oracle_gens(16)  # regression values (construction recalled)
    assert o.G[0].compress().hex() == "fc3b25801422672a6a8d3adb5d8457d4301fe92324b4fc56ae934c8713ddfe2d"
    assert o.G[1].compress().hex() == "ae817fdef62f713dd169dc8a26406f68be0bd3cd53652614636b0801567c4264"
    assert o.H[0].compress().hex() == "ba698f6dd08c501e32b55d2ee7259f6019d629fa2ba4d7039c5de157cba4df73"


def test_poseidon_constants_trap_T1():
    mds, rc = g._load_params_blob()
    assert len(mds) == 36 and len(rc) == 960
    assert mds[0] == 0x0b3022f895520ff40758ce8739d0b965997e8c819aa0d1cc3cafd96810b0e1b8
    with pytest.raises(ValueError):
        g.PoseidonParams(6, 4, 4, 200)   # not enough round constants (gadget_poseidon.rs:59-61)
    with pytest.raises(ValueError):
        g.PoseidonParams(5, 4, 4, 10)    # only width 6 (gadget_poseidon.rs:75-82)


def test_scalar_utils_semantics():
    assert g.get_bits(6, 4) == [0, 1, 1, 0]
    assert g.get_base_4_repr(18, 1) == [0, 1, 0, 2]
    assert g.get_base_4_repr(0, 2) == [0] * 8


@pytest.mark.parametrize("name", ["factors", "bound_check", "set_membership"])
def test_prove_verify_and_golden(name):
    import frontend_cases as fc
    gd = GOLD[name]
    _, _, _, sc, cap = fc.case(name, 0)
    ob = common.oracle_batch(lambda j: fc.case(name, j)[3], cap, 2)
    assert [p.hex() for p in ob["proofs"]] == gd["proofs"]
    assert (ob["n"], ob["q"], ob["m"]) == (gd["n"], gd["q"], gd["m"])
    obp = common.oracle_gens(cap)
    pf, comms = ob["proofs"][0], ob["comms"][0]
    assert sc.verify(common.PC, obp, pf, comms)
    # the negative tests the reference lacks
    bad = bytearray(pf); bad[len(bad) // 2] ^= 1
    with pytest.raises((VerificationError, R1CSError)):
        sc.verify(common.PC, obp, bytes(bad), comms)
    wrong = list(comms); wrong[0] = (BASEPOINT * 7).compress()
    with pytest.raises(VerificationError):
        sc.verify(common.PC, obp, pf, wrong)
    with pytest.raises(R1CSError):
        sc.verify(common.PC, obp, pf[:-1], comms)   # FormatError


def test_circuit_shapes_appendix_A():
    shapes = {k: (v["n"], v["q"], v["m"]) for k, v in GOLD.items()}
    assert shapes["factors"] == (1, 3, 2)
    assert shapes["bound_check_64"] == (128, 261, 3)
    assert shapes["poseidon_hash_2_cube"] == (376, 753, 6)
    assert shapes["poseidon_hash_2_inverse"] == (564, 1317, 6)
    assert shapes["poseidon_hash_4_inverse"] == (564, 1317, 6)
    assert shapes["set_membership"] == (21, 51, 8)
    assert shapes["vsmt_4_l4"] == (583 * 4, 1354 * 4 + 2, 3 * 4 + 4)
    assert len(bytes.fromhex(GOLD["bound_check_64"]["proofs"][0])) == 865
    assert len(bytes.fromhex(GOLD["poseidon_hash_2_cube"]["proofs"][0])) == 993
