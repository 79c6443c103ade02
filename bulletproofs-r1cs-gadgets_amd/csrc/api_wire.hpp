// C ABI: proof wire format.
#pragma once
#include "api_common.hpp"
// ---------------------------------------------------------------- proof wire format
// R1CSProof::to_bytes / from_bytes of the bulletproofs crate the reference depends on (Cargo.toml:22-26): a version byte
// (0 = one-phase: the phase-2 commitments are the identity and are not written; 1 = two-phase: 14 leading elements),
// 32-byte elements, the inner-product proof last.  from_bytes copies points undecoded and demands canonical scalars.
extern "C" int bpr1cs_proof_parse(const uint8_t* bytes, size_t len, bpr1cs_proof* out) {
    if (!bytes || !out) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (len < 1 || (len - 1) % 32 != 0) return BPR1CS_ERR_FORMAT;
    const uint8_t version = bytes[0];
    if (version > 1) return BPR1CS_ERR_FORMAT;
    const size_t k = (len - 1) / 32, lead = version ? 14 : 11;
    if (k < lead + 2 || ((k - lead - 2) & 1) != 0) return BPR1CS_ERR_FORMAT;
    const size_t lg = (k - lead - 2) / 2;
    if (lg >= 32) return BPR1CS_ERR_FORMAT;
    const uint8_t* el = bytes + 1;
    memset(out, 0, sizeof *out);
    auto take = [&](uint8_t* dst) { memcpy(dst, el, 32); el += 32; };
    take(out->A_I1); take(out->A_O1); take(out->S1);
    if (version) { take(out->A_I2); take(out->A_O2); take(out->S2); }
    take(out->T_1); take(out->T_3); take(out->T_4); take(out->T_5); take(out->T_6);
    take(out->t_x); take(out->t_x_blinding); take(out->e_blinding);
    out->lg_n = (uint32_t)lg;
    for (size_t i = 0; i < lg; i++) { take(out->L[i]); take(out->R[i]); }
    take(out->ipp_a); take(out->ipp_b);
    if (!host_scalar_canonical(out->t_x) || !host_scalar_canonical(out->t_x_blinding) || !host_scalar_canonical(out->e_blinding) ||
        !host_scalar_canonical(out->ipp_a) || !host_scalar_canonical(out->ipp_b))
        return BPR1CS_ERR_FORMAT;
    return BPR1CS_OK;
}
extern "C" size_t bpr1cs_proof_serialized_len(const bpr1cs_proof* p) {
    if (!p || p->lg_n >= 32) return 0;
    bool phase2 = false;
    for (int i = 0; i < 32; i++) phase2 = phase2 || p->A_I2[i] || p->A_O2[i] || p->S2[i];
    return 1 + 32 * ((phase2 ? 14 : 11) + 2 * (size_t)p->lg_n + 2);
}
extern "C" int bpr1cs_proof_serialize(const bpr1cs_proof* p, uint8_t* out, size_t cap, size_t* len_out) {
    if (!p || !out) return BPR1CS_ERR_INVALID_ARGUMENT;
    const size_t len = bpr1cs_proof_serialized_len(p);
    if (len == 0) return BPR1CS_ERR_FORMAT;
    if (cap < len) return BPR1CS_ERR_INVALID_ARGUMENT;
    const bool phase2 = len == 1 + 32 * (14 + 2 * (size_t)p->lg_n + 2);
    uint8_t* o = out;
    *o++ = phase2 ? 1 : 0;
    auto put = [&](const uint8_t* src) { memcpy(o, src, 32); o += 32; };
    put(p->A_I1); put(p->A_O1); put(p->S1);
    if (phase2) { put(p->A_I2); put(p->A_O2); put(p->S2); }
    put(p->T_1); put(p->T_3); put(p->T_4); put(p->T_5); put(p->T_6);
    put(p->t_x); put(p->t_x_blinding); put(p->e_blinding);
    for (uint32_t i = 0; i < p->lg_n; i++) { put(p->L[i]); put(p->R[i]); }
    put(p->ipp_a); put(p->ipp_b);
    if (len_out) *len_out = len;
    return BPR1CS_OK;
}
