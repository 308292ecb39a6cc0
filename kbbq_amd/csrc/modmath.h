// modmath.h -- exact h % d for a 32-bit hash h and a block count d (get_block, bloom.hh:99-105), shared by the
// device kernels and the host-side test hook.
//
// d <= 2^31: Barrett with a 32-bit magic M = floor(2^32 / d): q' = mulhi(h, M) is q or q - 1 (h*M/2^32 lies in
// (h/d - 1, h/d] because h < 2^32 and 2^32 mod d < d), so r' = h - q'*d lies in [0, 2d) -- below 2^32 -- and one
// conditional subtraction finishes it.  Two 32-bit multiplies instead of the seven of the 64-bit form, which
// remains for larger d (Lemire's fastmod: low = M64 * h, result = mulhi64(low, d)).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KBBQ_HD __host__ __device__ __forceinline__
#else
#define KBBQ_HD inline
#endif

namespace kbbq {

struct ModMagic {
    uint64_t d;      // the divisor (number of blocks)
    uint64_t m64;    // 2^64 / d + 1, 0 when d >= 2^32 (the hash is then its own remainder)
    uint32_t m32;    // floor(2^32 / d) when 2 <= d <= 2^31, else 0
};

inline ModMagic make_mod_magic(uint64_t d) {
    ModMagic m;
    m.d = d;
    m.m64 = d > 0xFFFFFFFFULL ? 0 : (~0ULL / d + 1);
    m.m32 = (d >= 2 && d <= 0x80000000ULL) ? (uint32_t)(0x100000000ULL / d) : 0;
    return m;
}

KBBQ_HD uint32_t mod_hash(uint32_t h, uint64_t d, uint64_t m64, uint32_t m32) {
    if (m32) {
        const uint32_t dd = (uint32_t)d;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t q = __umulhi(h, m32);
#else
        const uint32_t q = (uint32_t)(((uint64_t)h * m32) >> 32);
#endif
        uint32_t r = h - q * dd;
        if (r >= dd) r -= dd;
        return r;
    }
    if (d > 0xFFFFFFFFULL) return h;
    if (d == 1) return 0;
    const uint64_t low = m64 * h;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__umul64hi(low, d);
#else
    return (uint32_t)(((unsigned __int128)low * d) >> 64);
#endif
}

}  // namespace kbbq
