// pt_math.h -- FP64 math and the sample streams of the path-tracing core.
//
// Everything here is built from IEEE +,-,*,/ and sqrt only (compile with
// -ffp-contract=off), so the gfx950 kernels and the host-side set-up code of
// libptcore produce the same bits for the same input.  The routines follow the
// algorithms of the Go standard library that the reference engine calls
// (math.Sin/Cos at internal/engine/math.go:120-121, math.Tan at camera.go:26,
// math.Exp at renderer.go:361-363, math.Pow at materials.go:230, math.Min/Max at
// materials.go:185, math.go:49, renderer.go:378,384) so results track the
// reference to the last bit wherever the Go routine is the portable one.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace ptm {

PT_HD bool is_nan(double x) { return x != x; }

PT_HD double from_bits(uint64_t b) {
    union { uint64_t u; double d; } c;
    c.u = b;
    return c.d;
}
PT_HD uint64_t to_bits(double d) {
    union { uint64_t u; double d; } c;
    c.d = d;
    return c.u;
}
PT_HD bool sign_bit(double x) { return (to_bits(x) >> 63) != 0; }
PT_HD bool is_inf(double x) { return (to_bits(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
PT_HD double f_abs(double x) { return from_bits(to_bits(x) & 0x7fffffffffffffffULL); }

PT_HD double inf_pos() { return from_bits(0x7ff0000000000000ULL); }
PT_HD double qnan() { return from_bits(0x7ff8000000000001ULL); }
PT_HD double max_float64() { return from_bits(0x7fefffffffffffffULL); }

// IEEE square root.  On the device: the compiler expands __builtin_sqrt into v_rsq_f64 + two refinement steps wrapped in a range
// scaling (x < 2^-767 is scaled up by 2^256 and the root back down: a compare, two selects and two v_ldexp_f64 per root) and a
// fix-up for zeros and infinity (a class test and two selects).  For a positive normal x >= 2^-767 -- every root the kernels
// take, bar the exact zeros -- the scaling is by 2^0 and the fix-up does nothing: the same refinement without them gives the same
// bits (checked against __builtin_sqrt on 4 * 10^9 operands by pt_debug_div_selftest); everything else takes the builtin.
#if defined(__HIP_DEVICE_COMPILE__)
// (the compiler's sequence as a function of its own: inlined next to the short form it costs the trace kernel registers it does not
// have -- the stream state went to scratch memory across the cosine sampling -- while hardly any root ever comes here)
__device__ __attribute__((noinline)) inline double f_sqrt_any(double x) { return __builtin_sqrt(x); }
#endif
template <bool OUTLINE = true>  // false: the compiler's sequence inlined too (the BVH walk: a call inside its loop costs it more)
PT_HD double f_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t e = (uint32_t)(to_bits(x) >> 52);  // sign and exponent field
    if (e - 256u <= 0x7feu - 256u) {                    // +2^-767 <= x < +inf
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    }
    return OUTLINE ? f_sqrt_any(x) : __builtin_sqrt(x);
#else
    return __builtin_sqrt(x);
#endif
}

// math.Min / math.Max with Go's special cases (-Inf/+Inf win, NaN propagates,
// signed zeros ordered).
PT_HD double go_min(double x, double y) {
    if ((is_inf(x) && x < 0) || (is_inf(y) && y < 0)) return -inf_pos();
    if (is_nan(x) || is_nan(y)) return qnan();
    if (x == 0 && x == y) return sign_bit(x) ? x : y;
    return x < y ? x : y;
}
PT_HD double go_max(double x, double y) {
    if ((is_inf(x) && x > 0) || (is_inf(y) && y > 0)) return inf_pos();
    if (is_nan(x) || is_nan(y)) return qnan();
    if (x == 0 && x == y) return sign_bit(x) ? y : x;
    return x > y ? x : y;
}

// Cody-Waite split of pi/4 used by math.Sin, Cos and Tan.
#define PTM_PI4A 7.85398125648498535156e-1
#define PTM_PI4B 3.77489470793079817668e-8
#define PTM_PI4C 2.69515142907905952645e-15
#define PTM_4_OVER_PI 1.2732395447351628

PT_HD double sin_poly(double z, double zz) {
    return z + z * zz *
                   ((((((1.58962301576546568060e-10 * zz) + -2.50507477628578072866e-8) * zz + 2.75573136213857245213e-6) * zz +
                      -1.98412698295895385996e-4) * zz + 8.33333333332211858878e-3) * zz + -1.66666666666666307295e-1);
}
PT_HD double cos_poly(double zz) {
    return 1.0 - 0.5 * zz +
           zz * zz *
               ((((((-1.13585365213876817300e-11 * zz) + 2.08757008419747316778e-9) * zz + -2.75573141792967388112e-7) * zz +
                  2.48015872888517045348e-5) * zz + -1.38888888888730564116e-3) * zz + 4.16666666666665929218e-2);
}

// sin and cos of the same angle, 0 <= x < 2^29 (the engine only passes
// phi = 2*pi*r1 with r1 in [0,1)).  One range reduction serves both.
PT_HD void sincos_pos(double x, double *s, double *c) {
    uint32_t j = (uint32_t)(x * PTM_4_OVER_PI);
    double y = (double)j;
    if (j & 1u) { j++; y++; }
    j &= 7u;
    double z = ((x - y * PTM_PI4A) - y * PTM_PI4B) - y * PTM_PI4C;
    double zz = z * z;
    double ps = sin_poly(z, zz);
    double pc = cos_poly(zz);
    // sin: octants 1,2 use the cosine polynomial; sign flips for j > 3
    bool ssign = j > 3u;
    uint32_t js = ssign ? j - 4u : j;
    double sv = (js == 1u || js == 2u) ? pc : ps;
    // cos: octants 1,2 use the sine polynomial; sign flips for j > 3, again for (j mod 4) > 1
    bool csign = (j > 3u) != (js > 1u);
    double cv = (js == 1u || js == 2u) ? ps : pc;
    *s = ssign ? -sv : sv;
    *c = csign ? -cv : cv;
}

// math.Tan for 0 < x < 2^29 (camera set-up, host side).
PT_HD double tan_pos(double x) {
    uint64_t j = (uint64_t)(x * PTM_4_OVER_PI);
    double y = (double)j;
    if (j & 1u) { j++; y++; }
    double z = ((x - y * PTM_PI4A) - y * PTM_PI4B) - y * PTM_PI4C;
    double zz = z * z;
    if (zz > 1e-14)
        y = z + z * (zz * (((-1.30936939181383777646e4 * zz) + 1.15351664838587416140e6) * zz + -1.79565251976484877988e7) /
                     ((((zz + 1.36812963470692954678e4) * zz + -1.32089234440210967447e6) * zz + 2.50083801823357915839e7) * zz +
                      -5.38695755929454629881e7));
    else
        y = z;
    if (j & 2u) y = -1 / y;
    return y;
}
PT_HD double go_tan(double x) {
    if (x == 0 || is_nan(x)) return x;
    if (is_inf(x)) return qnan();
    bool neg = x < 0;
    if (neg) x = -x;
    if (x >= 536870912.0) return qnan();
    double y = tan_pos(x);
    return neg ? -y : y;
}

// y * 2^k for a normal result (k small): exact scaling through the exponent field.
PT_HD double scale_pow2(double y, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(y, k);
#else
    return __builtin_ldexp(y, k);
#endif
}

// math.Exp, portable Go routine (argument reduction by ln2 hi/lo + degree-5 rational).
PT_HD double go_exp(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01;
    const double Ln2Lo = 1.90821492927058770002e-10;
    const double Log2e = 1.44269504088896338700e+00;
    if (is_nan(x)) return x;
    if (is_inf(x)) return x > 0 ? x : 0.0;
    if (x > 7.09782712893383973096e+02) return inf_pos();
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double NearZero = 1.0 / (double)(1 << 28);
    if (-NearZero < x && x < NearZero) return 1 + x;
    int k = 0;
    if (x < 0) k = (int)(Log2e * x - 0.5);
    else if (x > 0) k = (int)(Log2e * x + 0.5);
    double hi = x - (double)k * Ln2Hi;
    double lo = (double)k * Ln2Lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (1.66666666666666657415e-01 +
                        t * (-2.77777777770155933842e-03 +
                             t * (6.61375632143793436117e-05 + t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return scale_pow2(y, k);
}

// math.Pow(x, 5) for x >= 0: Go multiplies mantissas (x, x^2, x^4 by repeated
// squaring; result x * x^4) and adds exponents, which rounds like the plain
// products below whenever no intermediate is subnormal -- true for every
// x = 1 - cos(theta) the engine can produce (x is 0 or >= 2^-53).
PT_HD double go_pow5(double x) {
    if (x == 1) return 1;
    if (is_nan(x)) return qnan();
    if (x == 0) return 0;
    double x2 = x * x;
    double x4 = x2 * x2;
    return x * x4;
}

// ---------------------------------------------------------------- sample streams
// One stream per (seed, pixel, sample), honouring the Float64 contract of internal/engine/random.go:27-34 (uniform
// multiples of 2^-53 in [0,1)) and nothing else of the reference's generator, which is seeded from the clock
// (random.go:14-16) and so has no stream to reproduce.
//   key    h = mix64(mix64(seed + G) + (pixel << 32 | sample))      pixel < 2^28, sample < 2^31: one 64-bit word, one
//              splitmix64 finaliser (a bijection: distinct (pixel, sample) give distinct h)
//   state  MWC64X (D. B. Thomas, "The MWC64X Random Number Generator", 2011): x = low word of h, carry
//              c = (high word >> 1) + 1, so that 1 <= c <= 2^31 < A: never one of the two fixed points (0, 0), (2^32-1, A-1).
//              The shift drops bit 32 of h: the state is a 63-bit function of the 64-bit key, so two (pixel, sample) whose
//              hashes differ in that bit alone share a stream -- about N^2 / 2^64 such pairs among N streams (~60 in the
//              3.4 * 10^10 of a C5 frame): distinct streams are overwhelmingly likely, not guaranteed.
//   step   out = x ^ c;  (c, x) <- A * x + c   with A = 4294883355; A * 2^32 - 1 is a safe prime, every other state
//              lies on one of two cycles of A * 2^31 - 1 ~ 2^63 steps.  One v_mad_u64_u32 and one v_xor_b32 on gfx950.
//   draw   two steps: (out1 << 21 | out2 >> 11) * 2^-53.
// Round 3 replaced the splitmix64 stream (one 64-bit finaliser per draw: 23 vector instructions, 21 of them at 4 cycles per
// wave) by this one (10 instructions, 7 at 4 cycles); the CPU checker, the golden fixtures and the kernels changed together.
#define PTM_GOLDEN 0x9E3779B97F4A7C15ULL
#define PTM_MWC_A 4294883355ULL

PT_HD uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
PT_HD uint64_t seed_key(uint64_t seed) { return mix64(seed + PTM_GOLDEN); }
PT_HD uint64_t stream_init(uint64_t seed_key_, uint64_t pixel, uint64_t sample) {
    const uint64_t h = mix64(seed_key_ + ((pixel << 32) | (sample & 0xffffffffULL)));
    const uint32_t x = (uint32_t)h, c = ((uint32_t)(h >> 32) >> 1) + 1u;
    return ((uint64_t)c << 32) | x;
}
PT_HD double stream_next(uint64_t &state) {
    uint32_t x = (uint32_t)state, c = (uint32_t)(state >> 32);
    const uint32_t o1 = x ^ c;
    uint64_t t = (uint64_t)x * PTM_MWC_A + c;
    x = (uint32_t)t;
    c = (uint32_t)(t >> 32);
    const uint32_t o2 = x ^ c;
    state = (uint64_t)x * PTM_MWC_A + c;
    // (o1 * 2^21 + (o2 >> 11)) * 2^-53, every step exact: two conversions, one scaling, one fma
    return __builtin_fma((double)o1, 0x1p-32, (double)(o2 & 0xfffff800u) * 0x1p-64);
}

}  // namespace ptm
