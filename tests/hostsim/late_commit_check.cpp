// The C++ Prover's chain-ahead (host/r1cs.hpp ChainAhead) against the path without it, byte for byte, on a Prover driven by hand:
// (1) commitments first, then the constraint system - the reference's order: prove() goes through bpr1cs_prove_batch_draws;
// (2) a commit() AFTER the first multiplier - legal, never done by the reference: the chain started too early is dropped and prove()
//     takes bpr1cs_prove_batch_transcripts; (3) a transcript that already holds a message.  Linked against the CPU simulator.
#include <cstdio>
#include "../../bulletproofs-r1cs-gadgets_amd/host/r1cs.hpp"
using namespace bpr1cs;

static std::vector<uint8_t> run(bool ahead, bool late, bool advanced, bool* used_chain) {
    BulletproofGens bp(8, 1);
    PedersenGens pc(bp);
    Transcript t("late-commit");
    if (advanced) t.append_message("ctx", (const uint8_t*)"session", 7);
    Prover p(pc, t);
    p.chain_ahead_enabled = ahead;
    p.chain_ahead_min_commitments = 0;
    std::array<uint8_t, 32> seed;
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)(7 * i + 1);
    p.set_rng_seed(seed);
    const Scalar x(6), y(7), z(42), w(5);
    auto cx = p.commit(x, Scalar(11));
    auto cy = p.commit(y, Scalar(12));
    Variable vz = Variable::One();
    if (!late) vz = p.commit(z, Scalar(13)).second;
    MulVars m1 = p.multiply(LinearCombination(cx.second), LinearCombination(cy.second));   // x * y   (the synthesis begins here)
    if (late) vz = p.commit(z, Scalar(13)).second;                                          // a commitment after it
    p.constrain(LinearCombination(m1.out) - LinearCombination(vz));                         // = z
    MulVars m2 = p.multiply(LinearCombination(m1.out) - LinearCombination(Scalar(37)), LinearCombination(cx.second) - LinearCombination(Scalar(1)));   // (42-37)*(6-1)
    p.constrain(LinearCombination(m2.out) - LinearCombination(Scalar(25)));
    (void)w;
    const bool had = (bool)p.chain;
    const size_t chain_m = had ? p.chain->m : 0;
    R1CSProof pf = p.prove(bp);
    if (used_chain) *used_chain = had && chain_m == 3;
    std::vector<uint8_t> out = pf.to_bytes();
    uint8_t ch[16];
    t.challenge_bytes("after", ch, 16);   // the caller's transcript is where upstream's is after prove()
    out.insert(out.end(), ch, ch + 16);
    for (auto& c : {cx.first, cy.first}) { auto b = c.to_bytes(); out.insert(out.end(), b.begin(), b.end()); }
    return out;
}

int main() {
    int bad = 0;
    for (int advanced = 0; advanced < 2; advanced++)
        for (int late = 0; late < 2; late++) {
            bool used = false;
            std::vector<uint8_t> a = run(true, late, advanced, &used), b = run(false, late, advanced, nullptr);
            const bool same = a == b, expect_used = !late;
            printf("advanced=%d late=%d: %s, chain-ahead %s\n", advanced, late, same ? "same bytes" : "BYTES DIFFER", used ? "used" : "not used");
            if (!same || used != expect_used) bad++;
        }
    printf("%s\n", bad ? "FAIL" : "OK");
    return bad;
}
