#!/usr/bin/env python3
"""Materialise the *effective* Poseidon parameters of the reference as data.

Reads the 996 hex literals of /root/reference/src/poseidon_constants.rs
(MDS_ENTRIES :1, ROUND_CONSTS :10) and applies the reference's own parsing
semantics (src/scalar_utils.rs:232-237: hex text -> bytes in text order ->
Scalar::from_bytes_mod_order, i.e. little-endian, reduced mod l; SURVEY trap
T1), then writes the resulting canonical 32-byte scalars:

    36  MDS entries  (row-major, MDS[i][j] at 6*i+j)
    960 round keys

Output (identical bytes, two consumers that must not share a file path):
    tests/golden/poseidon_params_ristretto.bin            (oracle / tests)
    bulletproofs-r1cs-gadgets_amd/data/poseidon_params_ristretto.bin (product)

Runs only in the build container (needs /root/reference); the .bin is data
(parsed values), not reference source text.
"""
import os
import re
import sys

L = 2**252 + 27742317777372353535851937790883648493
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = open("/root/reference/src/poseidon_constants.rs").read()
    hexes = re.findall(r'"(0x[0-9a-fA-F]{64})"', src)
    assert len(hexes) == 36 + 960, len(hexes)
    blob = bytearray()
    for h in hexes:
        raw = bytes.fromhex(h[2:])
        blob += (int.from_bytes(raw, "little") % L).to_bytes(32, "little")
    for rel in ("tests/golden/poseidon_params_ristretto.bin",
                "bulletproofs-r1cs-gadgets_amd/data/poseidon_params_ristretto.bin"):
        path = os.path.join(ROOT, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(blob)
        print("wrote", rel, len(blob))
    # sanity anchors from SURVEY §8c(5)
    assert int(hexes[0], 16) * 6 % L == 1
    assert blob[:32][::-1].hex() == "0b3022f895520ff40758ce8739d0b965997e8c819aa0d1cc3cafd96810b0e1b8"


if __name__ == "__main__":
    sys.exit(main())
