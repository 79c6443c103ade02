// The limb-per-lane power chain (csrc/fe_wide.hpp) against the lane's own fe_pow22523, and the wavefront form of compress against
// ge_compress, on the device: random elements of every limb class the callers produce (canonical bytes, centred results of fe_mul,
// sums and differences), the corner values, and random points.  Prints "mismatches 0" when every canonical encoding agrees;
// tests/test_gpu_wide.py builds and runs it.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibulletproofs-r1cs-gadgets_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fe.hpp"
#include "sc.hpp"
#include "ge.hpp"
#include "fe_wide.hpp"
struct fe_pow_wave {
    __device__ fe operator()(const fe& z) const { return fe_pow22523_wave(z); }
};
__device__ inline uint32_t xs(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// one wavefront per workgroup; workgroup w checks `per` values
__global__ void __launch_bounds__(64) k_check(uint32_t seed, int per, uint32_t* bad, uint64_t* cyc) {
    uint32_t s = seed + 7919u * blockIdx.x;   // uniform over the wavefront: every lane builds the same values
    uint32_t nbad = 0;
    uint64_t tw = 0, tl = 0;
    for (int it = 0; it < per; it++) {
        fe z;
        const uint32_t cls = xs(s) % 6u;
        for (int k = 0; k < 9; k++) {
            const uint32_t r = xs(s);
            if (cls == 0) z.v[k] = (int32_t)(r & FE_MASK);                                     // canonical-byte class [0, 2^29)
            else if (cls == 1) z.v[k] = (int32_t)(r % ((1u << 29) + (1u << 24))) - (1 << 28) - (1 << 23);   // centred class N
            else if (cls == 2) z.v[k] = (int32_t)(r & 3u) - 1;                                   // tiny
            else if (cls == 3) z.v[k] = (r & 1u) ? (1 << 28) + (1 << 23) : -(1 << 28) - (1 << 23);      // the bounds of class N
            else if (cls == 4) z.v[k] = k == (int)(r % 9u) ? 1 : 0;                              // a single limb
            else z.v[k] = (int32_t)(r % (1u << 30)) - (1 << 29);                                // 2N (fe_carry brings it back)
        }
        const uint64_t a0 = clock64();
        const fe w = fe_pow22523_wave(z);
        const uint64_t a1 = clock64();
        const fe l = fe_pow22523(z);
        const uint64_t a2 = clock64();
        tw += a1 - a0; tl += a2 - a1;
        uint8_t bw[32], bl[32];
        fe_tobytes(w, bw); fe_tobytes(l, bl);
        int diff = 0;
        for (int i = 0; i < 32; i++) diff |= bw[i] ^ bl[i];
        nbad += diff != 0;
        // a point: a multiple of the base point with a random projective factor; lanes get DIFFERENT factors so that only lane 0's matters
        ge p = ge_basepoint();
        const uint32_t steps = xs(s) % 7u;
        for (uint32_t j = 0; j < steps; j++) p = ge_add_ge(p, j & 1 ? p : ge_basepoint());
        fe lam = z;
        lam.v[0] += (int32_t)threadIdx.x + 1;
        lam = fe_carry(lam);
        p.X = fe_mul(p.X, lam); p.Y = fe_mul(p.Y, lam); p.Z = fe_mul(p.Z, lam); p.T = fe_mul(p.T, lam);
        uint8_t e0[32], e1[32];
        ge_compress(p, e0);
        ge_compress_t(p, e1, fe_pow_wave{});
        diff = 0;
        for (int i = 0; i < 32; i++) diff |= e0[i] ^ e1[i];
        if (threadIdx.x == 0) nbad += diff != 0;
    }
    if (threadIdx.x == 0) { atomicAdd(bad, nbad); if (blockIdx.x == 0) { cyc[0] = tw / per; cyc[1] = tl / per; } }
}
int main() {
    uint32_t* bad; uint64_t* cyc;
    if (hipMalloc(&bad, 4) != hipSuccess || hipMalloc(&cyc, 16) != hipSuccess) { printf("no device\n"); return 2; }
    hipMemset(bad, 0, 4);
    const int wgs = 256, per = 24;
    hipLaunchKernelGGL(k_check, dim3(wgs), dim3(64), 0, 0, 0x2545f491u, per, bad, cyc);
    uint32_t h = 1; uint64_t c[2] = {0, 0};
    if (hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("device error\n"); return 2; }
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("values %d\nmismatches %u\nticks per power: wavefront %llu, lane %llu\n", wgs * per * 2, h, (unsigned long long)c[0], (unsigned long long)c[1]);
    return h != 0;
}
