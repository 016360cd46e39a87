/*
 * pt_oracle.c -- TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 * Plain-C FP64 restatement of the reference Go CPU path tracer
 * (/root/reference/internal/engine).  Every function cites the Go file:line it
 * follows.  Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile);
 * Go on amd64 (GOAMD64=v1) never fuses multiply-add, so contraction must be off.
 *
 * Deliberate differences from the reference, both forced by it:
 *   * RNG: the reference seeds math/rand from the wall clock per worker
 *     (random.go:14-16), so it has no reproducible stream.  Here every
 *     (seed, pixel, sample) owns its own MWC64X stream (keyed by a splitmix64 hash) honouring the
 *     Float64 contract of random.go:27-34 (53-bit uniform in [0,1)).  Draw ORDER
 *     inside a sample is the reference's.
 *   * math.Exp: Go/amd64 uses an assembly routine; this follows the portable Go
 *     routine (same <1 ulp class).  Sin/Cos/Tan/Pow follow the portable Go
 *     routines, which are what amd64 runs.
 * PARITY UNPINNED by the reference (no tests / vectors exist there).
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------ */
/* Go math package restatements (stdlib go1.25, not vendored)          */
/* ------------------------------------------------------------------ */

static int is_nan(double x) { return x != x; }

/* math.Min: -Inf wins, NaN propagates, -0 < +0 */
double ora_min(double x, double y) {
    if ((isinf(x) && x < 0) || (isinf(y) && y < 0)) return -INFINITY;
    if (is_nan(x) || is_nan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? x : y;
    return x < y ? x : y;
}

/* math.Max: +Inf wins, NaN propagates, +0 > -0 */
double ora_max(double x, double y) {
    if ((isinf(x) && x > 0) || (isinf(y) && y > 0)) return INFINITY;
    if (is_nan(x) || is_nan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}

/* Cephes constants shared by math.Sin / Cos / Tan (Go src/math/sin.go, tan.go) */
static const double PI4A = 7.85398125648498535156e-1;
static const double PI4B = 3.77489470793079817668e-8;
static const double PI4C = 2.69515142907905952645e-15;
static const double FOUR_OVER_PI = 1.2732395447351628; /* Go folds the constant 4/Pi exactly, then rounds */

static const double SINCOF[6] = {
    1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
    -1.98412698295895385996e-4, 8.33333333332211858878e-3,  -1.66666666666666307295e-1,
};
static const double COSCOF[6] = {
    -1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
    2.48015872888517045348e-5,   -1.38888888888730564116e-3, 4.16666666666665929218e-2,
};

static double poly_sin(double z, double zz) {
    return z + z * zz * ((((((SINCOF[0] * zz) + SINCOF[1]) * zz + SINCOF[2]) * zz + SINCOF[3]) * zz + SINCOF[4]) * zz + SINCOF[5]);
}
static double poly_cos(double zz) {
    return 1.0 - 0.5 * zz + zz * zz * ((((((COSCOF[0] * zz) + COSCOF[1]) * zz + COSCOF[2]) * zz + COSCOF[3]) * zz + COSCOF[4]) * zz + COSCOF[5]);
}

/* Arguments here stay below 2^29 (phi < 2*pi, theta/2 < pi), so Go's Payne-Hanek
 * branch (trigReduce) is never taken; larger inputs return NaN to make misuse loud. */
#define TRIG_REDUCE_THRESHOLD 536870912.0

double ora_sin(double x) {
    if (x == 0 || is_nan(x)) return x;
    if (isinf(x)) return NAN;
    int sign = 0;
    if (x < 0) { x = -x; sign = 1; }
    if (x >= TRIG_REDUCE_THRESHOLD) return NAN;
    uint64_t j = (uint64_t)(x * FOUR_OVER_PI);
    double y = (double)j;
    if (j & 1) { j++; y++; }
    j &= 7;
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) { sign = !sign; j -= 4; }
    double zz = z * z;
    if (j == 1 || j == 2) y = poly_cos(zz);
    else y = poly_sin(z, zz);
    return sign ? -y : y;
}

double ora_cos(double x) {
    if (is_nan(x)) return x;
    if (isinf(x)) return NAN;
    int sign = 0;
    if (x < 0) x = -x;
    if (x >= TRIG_REDUCE_THRESHOLD) return NAN;
    uint64_t j = (uint64_t)(x * FOUR_OVER_PI);
    double y = (double)j;
    if (j & 1) { j++; y++; }
    j &= 7;
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) { j -= 4; sign = !sign; }
    if (j > 1) sign = !sign;
    double zz = z * z;
    if (j == 1 || j == 2) y = poly_sin(z, zz);
    else y = poly_cos(zz);
    return sign ? -y : y;
}

static const double TANP[3] = {-1.30936939181383777646e4, 1.15351664838587416140e6, -1.79565251976484877988e7};
static const double TANQ[5] = {1.0, 1.36812963470692954678e4, -1.32089234440210967447e6, 2.50083801823357915839e7, -5.38695755929454629881e7};

double ora_tan(double x) {
    if (x == 0 || is_nan(x)) return x;
    if (isinf(x)) return NAN;
    int sign = 0;
    if (x < 0) { x = -x; sign = 1; }
    if (x >= TRIG_REDUCE_THRESHOLD) return NAN;
    uint64_t j = (uint64_t)(x * FOUR_OVER_PI);
    double y = (double)j;
    if (j & 1) { j++; y++; }
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    double zz = z * z;
    if (zz > 1e-14)
        y = z + z * (zz * (((TANP[0] * zz) + TANP[1]) * zz + TANP[2]) / ((((zz + TANQ[1]) * zz + TANQ[2]) * zz + TANQ[3]) * zz + TANQ[4]));
    else
        y = z;
    if (j & 2) y = -1 / y;
    return sign ? -y : y;
}

/* math.Exp portable routine (Go src/math/exp.go: exp + expmulti) */
double ora_exp(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01;
    const double Ln2Lo = 1.90821492927058770002e-10;
    const double Log2e = 1.44269504088896338700e+00;
    const double Overflow = 7.09782712893383973096e+02;
    const double Underflow = -7.45133219101941108420e+02;
    const double NearZero = 1.0 / (double)(1 << 28);
    const double P1 = 1.66666666666666657415e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (is_nan(x) || (isinf(x) && x > 0)) return x;
    if (isinf(x)) return 0;
    if (x > Overflow) return INFINITY;
    if (x < Underflow) return 0;
    if (-NearZero < x && x < NearZero) return 1 + x;
    int k = 0;
    if (x < 0) k = (int)(Log2e * x - 0.5);
    else if (x > 0) k = (int)(Log2e * x + 0.5);
    double hi = x - (double)k * Ln2Hi;
    double lo = (double)k * Ln2Lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return ldexp(y, k);
}

/* math.Pow (Go src/math/pow.go) for the integral, non-negative exponents the
 * engine uses (materials.go:230 calls Pow(1-cosine, 5)): the yi loop of pow(). */
double ora_pow(double x, double y) {
    if (y == 0 || x == 1) return 1;
    if (y == 1) return x;
    if (is_nan(x) || is_nan(y)) return NAN;
    if (x == 0) return 0; /* y > 0 here */
    if (isinf(x)) return x > 0 ? INFINITY : (fmod(y, 2) == 1 ? -INFINITY : INFINITY);
    double yi = floor(y);
    if (yi != y || y < 0) return NAN; /* not needed by the engine */
    double a1 = 1.0;
    int ae = 0;
    int xe;
    double x1 = frexp(x, &xe);
    for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 12) || (1 << 12) < xe) {
            ae += xe;
            break;
        }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe *= 2; /* Go: xe <<= 1 (defined for negative values there) */
        if (x1 < .5) { x1 += x1; xe--; }
    }
    return ldexp(a1, ae);
}

/* ------------------------------------------------------------------ */
/* RNG: counter-based stream, random.go:27-34 contract                 */
/* ------------------------------------------------------------------ */

#define GOLDEN 0x9E3779B97F4A7C15ULL
#define MWC_A 4294883355ULL /* MWC64X multiplier (D. B. Thomas 2011); MWC_A * 2^32 - 1 is a safe prime */

static uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

/* key = one splitmix64 finaliser over (seed key + (pixel << 32 | sample)); the MWC64X state is (x, c) = (low word,
 * (high word >> 1) + 1): 1 <= c <= 2^31 < MWC_A keeps it off the generator's two fixed points. */
uint64_t ora_stream_init(uint64_t seed, uint64_t pixel, uint64_t sample) {
    uint64_t h = mix64(mix64(seed + GOLDEN) + ((pixel << 32) | (sample & 0xffffffffULL)));
    uint32_t x = (uint32_t)h, c = ((uint32_t)(h >> 32) >> 1) + 1u;
    return ((uint64_t)c << 32) | x;
}

/* one MWC64X step: returns x ^ c, then (c, x) <- MWC_A * x + c */
static uint32_t mwc64x(uint64_t *state) {
    uint32_t x = (uint32_t)*state, c = (uint32_t)(*state >> 32);
    *state = (uint64_t)x * MWC_A + c;
    return x ^ c;
}

/* Float64 contract of random.go:27-34: a uniform multiple of 2^-53 in [0,1); 32 bits of one step over 21 of the next */
double ora_stream_next(uint64_t *state) {
    uint64_t hi = mwc64x(state);
    uint64_t lo = mwc64x(state);
    return (double)((hi << 21) | (lo >> 11)) * (1.0 / 9007199254740992.0);
}

typedef struct {
    uint64_t state;
    uint32_t draws;
} randSource;

static double Float64(randSource *rs) {
    rs->draws++;
    return ora_stream_next(&rs->state);
}

/* ------------------------------------------------------------------ */
/* math.go                                                             */
/* ------------------------------------------------------------------ */

typedef struct { double x, y, z; } vec3;

static vec3 v(double x, double y, double z) { vec3 r = {x, y, z}; return r; }
static vec3 vadd(vec3 a, vec3 b) { return v(a.x + b.x, a.y + b.y, a.z + b.z); }       /* math.go:11 */
static vec3 vsub(vec3 a, vec3 b) { return v(a.x - b.x, a.y - b.y, a.z - b.z); }       /* math.go:12 */
static vec3 vmul(vec3 a, double t) { return v(a.x * t, a.y * t, a.z * t); }           /* math.go:13 */
static vec3 vdiv(vec3 a, double t) { double invT = 1.0 / t; return v(a.x * invT, a.y * invT, a.z * invT); } /* math.go:14-17 */
static double vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      /* math.go:19 */
static vec3 vcross(vec3 a, vec3 b) {                                                  /* math.go:21-27 */
    return v(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static double vlength(vec3 a) { return sqrt(vdot(a, a)); }                            /* math.go:29 */
static vec3 vunit(vec3 a) {                                                           /* math.go:31-37 */
    double l = vlength(a);
    if (l == 0) return a;
    return vdiv(a, l);
}

/* math.go:39-46 */
static vec3 reflectVec(vec3 vv, vec3 n) {
    double dot = vdot(vv, n);
    return v(vv.x - n.x * 2 * dot, vv.y - n.y * 2 * dot, vv.z - n.z * 2 * dot);
}

/* math.go:48-64 */
static vec3 refractVec(vec3 uv, vec3 n, double etaiOverEtat) {
    double cosTheta = ora_min(-uv.x * n.x - uv.y * n.y - uv.z * n.z, 1.0);
    double px = uv.x + n.x * cosTheta;
    double py = uv.y + n.y * cosTheta;
    double pz = uv.z + n.z * cosTheta;
    px *= etaiOverEtat;
    py *= etaiOverEtat;
    pz *= etaiOverEtat;
    double perpLenSq = px * px + py * py + pz * pz;
    double par = -sqrt(fabs(1.0 - perpLenSq));
    return v(px + n.x * par, py + n.y * par, pz + n.z * par);
}

/* math.go:66-85 */
static vec3 randomInUnitSphere(randSource *rng) {
    for (;;) {
        double x = Float64(rng) * 2 - 1;
        double y = Float64(rng) * 2 - 1;
        double z = Float64(rng) * 2 - 1;
        double lenSq = x * x + y * y + z * z;
        if (lenSq >= 1.0) continue;
        return v(x, y, z);
    }
}

/* math.go:94-131 */
static vec3 randomCosineDirection(vec3 normal, randSource *rng) {
    double r1 = Float64(rng);
    double r2 = Float64(rng);
    double phi = 6.283185307179586 * r1; /* Go folds 2.0*math.Pi first */
    double cosTheta = sqrt(r2);
    double sinTheta = sqrt(1.0 - r2);
    vec3 u = fabs(normal.x) > 0.9 ? v(0, 1, 0) : v(1, 0, 0);
    vec3 w = normal;
    vec3 vVec = vunit(vcross(w, u));
    vec3 uVec = vcross(vVec, w);
    vec3 local = v(sinTheta * ora_cos(phi), sinTheta * ora_sin(phi), cosTheta);
    return v(local.x * uVec.x + local.y * vVec.x + local.z * w.x,
             local.x * uVec.y + local.y * vVec.y + local.z * w.y,
             local.x * uVec.z + local.y * vVec.z + local.z * w.z);
}

typedef struct { vec3 orig, dir; } ray;

/* ------------------------------------------------------------------ */
/* materials.go                                                        */
/* ------------------------------------------------------------------ */

enum { matLambert = 0, matMetal, matDielectric, matEmissive, matMirror };

typedef struct {
    int typ;
    vec3 albedo;
    double rough, ior;
    vec3 emit, absorption;
} material;

static double clampf(double x, double lo, double hi) { /* materials.go:57-65 */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* materials.go:28-55 */
static material convertMaterial(const ora_material *m) {
    material r;
    memset(&r, 0, sizeof r);
    vec3 al = v(m->albedo[0], m->albedo[1], m->albedo[2]);
    vec3 em = v(m->emit[0] * m->power, m->emit[1] * m->power, m->emit[2] * m->power);
    vec3 ab = v(m->absorption[0], m->absorption[1], m->absorption[2]);
    switch (m->type) {
    case 1: {
        double rough = m->rough;
        if (m->smoothness > 0) rough = 1.0 - clampf(m->smoothness, 0, 1);
        r.typ = matMetal; r.albedo = al; r.rough = clampf(rough, 0, 1);
        return r;
    }
    case 2: {
        double ior = m->ior;
        if (ior == 0) ior = 1.5;
        r.typ = matDielectric; r.albedo = al; r.ior = ior; r.absorption = ab;
        return r;
    }
    case 3:
        r.typ = matEmissive; r.emit = em;
        return r;
    case 4:
        r.typ = matMirror; r.albedo = al;
        return r;
    default:
        r.typ = matLambert; r.albedo = al; r.rough = clampf(m->rough, 0, 1);
        return r;
    }
}

typedef struct {
    vec3 p, normal;
    double t;
    int frontFace;
    material mat;
} hitRecord;

/* materials.go:67-72 */
static vec3 emitted(const material *m) {
    if (m->typ == matEmissive) return m->emit;
    return v(0, 0, 0);
}

/* materials.go:226-231 */
static double reflectance(double cosine, double refIdx) {
    double r0 = (1 - refIdx) / (1 + refIdx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * ora_pow(1 - cosine, 5);
}

/* materials.go:74-224.  Returns ok; writes attenuation + scattered. */
static int scatter(const material *m, randSource *rng, ray rIn, const hitRecord *rec, vec3 *attenuation, ray *scattered) {
    switch (m->typ) {
    case matLambert: { /* :76-97 */
        vec3 d = randomCosineDirection(rec->normal, rng);
        if (m->rough > 1e-6) {
            vec3 off = randomInUnitSphere(rng);
            d.x += off.x * m->rough * 0.1;
            d.y += off.y * m->rough * 0.1;
            d.z += off.z * m->rough * 0.1;
            d = vunit(d);
        }
        scattered->orig = rec->p;
        scattered->dir = d;
        *attenuation = m->albedo;
        return 1;
    }
    case matMetal: { /* :99-160 */
        double dirLen = sqrt(rIn.dir.x * rIn.dir.x + rIn.dir.y * rIn.dir.y + rIn.dir.z * rIn.dir.z);
        if (dirLen == 0) {
            *attenuation = v(0, 0, 0);
            scattered->orig = rec->p; scattered->dir = rIn.dir;
            return 0;
        }
        double invLen = 1.0 / dirLen;
        vec3 unitDir = v(rIn.dir.x * invLen, rIn.dir.y * invLen, rIn.dir.z * invLen);
        vec3 reflected = reflectVec(unitDir, rec->normal);
        if (m->rough > 1e-6) {
            vec3 sd = randomCosineDirection(reflected, rng);
            double alpha = m->rough * m->rough;
            double sx = reflected.x * (1.0 - alpha) + sd.x * alpha;
            double sy = reflected.y * (1.0 - alpha) + sd.y * alpha;
            double sz = reflected.z * (1.0 - alpha) + sd.z * alpha;
            double lenSq = sx * sx + sy * sy + sz * sz;
            if (lenSq < 1e-8) {
                sx = reflected.x; sy = reflected.y; sz = reflected.z;
            } else {
                double len = sqrt(lenSq);
                double inv = 1.0 / len;
                sx *= inv; sy *= inv; sz *= inv;
            }
            double dot = sx * rec->normal.x + sy * rec->normal.y + sz * rec->normal.z;
            if (dot <= 0) { sx = reflected.x; sy = reflected.y; sz = reflected.z; }
            *attenuation = m->albedo;
            scattered->orig = rec->p; scattered->dir = v(sx, sy, sz);
            return 1;
        }
        *attenuation = m->albedo;
        scattered->orig = rec->p; scattered->dir = reflected;
        return 1;
    }
    case matDielectric: { /* :162-200 */
        *attenuation = v(1, 1, 1);
        double ratio = rec->frontFace ? 1.0 / m->ior : m->ior;
        double dirLen = sqrt(rIn.dir.x * rIn.dir.x + rIn.dir.y * rIn.dir.y + rIn.dir.z * rIn.dir.z);
        if (dirLen == 0) {
            scattered->orig = rec->p; scattered->dir = rIn.dir;
            return 0;
        }
        double invLen = 1.0 / dirLen;
        vec3 unitDir = v(rIn.dir.x * invLen, rIn.dir.y * invLen, rIn.dir.z * invLen);
        double cosTheta = ora_min(-(unitDir.x * rec->normal.x + unitDir.y * rec->normal.y + unitDir.z * rec->normal.z), 1.0);
        double sinTheta = sqrt(1.0 - cosTheta * cosTheta);
        int cannotRefract = ratio * sinTheta > 1.0;
        double reflectProb = reflectance(cosTheta, ratio);
        vec3 direction;
        /* Go's || short-circuits: no draw under total internal reflection (:193) */
        if (cannotRefract || reflectProb > Float64(rng)) direction = reflectVec(unitDir, rec->normal);
        else direction = refractVec(unitDir, rec->normal, ratio);
        scattered->orig = rec->p; scattered->dir = direction;
        return 1;
    }
    case matEmissive: /* :202-203 */
        *attenuation = v(0, 0, 0);
        memset(scattered, 0, sizeof *scattered);
        return 0;
    case matMirror: { /* :205-221 */
        double dirLen = sqrt(rIn.dir.x * rIn.dir.x + rIn.dir.y * rIn.dir.y + rIn.dir.z * rIn.dir.z);
        if (dirLen == 0) {
            *attenuation = v(0, 0, 0);
            scattered->orig = rec->p; scattered->dir = rIn.dir;
            return 0;
        }
        double invLen = 1.0 / dirLen;
        vec3 unitDir = v(rIn.dir.x * invLen, rIn.dir.y * invLen, rIn.dir.z * invLen);
        scattered->orig = rec->p; scattered->dir = reflectVec(unitDir, rec->normal);
        *attenuation = m->albedo;
        return 1;
    }
    }
    *attenuation = v(0, 0, 0);
    memset(scattered, 0, sizeof *scattered);
    return 0;
}

/* ------------------------------------------------------------------ */
/* objects.go                                                          */
/* ------------------------------------------------------------------ */

enum { kSphere = 0, kPlane = 1, kBox = 2 };

typedef struct {
    int kind;
    vec3 a;        /* sphere center | plane point | box min */
    vec3 b;        /* -             | plane normal | box max */
    double radius;
    material mat;
} hittable;

/* objects.go:37-89 */
static int sphere_hit(const hittable *s, ray r, double tMin, double tMax, hitRecord *rec) {
    double ocX = r.orig.x - s->a.x, ocY = r.orig.y - s->a.y, ocZ = r.orig.z - s->a.z;
    double a = r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z;
    double halfB = ocX * r.dir.x + ocY * r.dir.y + ocZ * r.dir.z;
    double ocLenSq = ocX * ocX + ocY * ocY + ocZ * ocZ;
    double radiusSq = s->radius * s->radius;
    double c = ocLenSq - radiusSq;
    double discriminant = halfB * halfB - a * c;
    if (discriminant < 0) return 0;
    double sqrtD = sqrt(discriminant);
    double root = (-halfB - sqrtD) / a;
    if (root < tMin || root > tMax) {
        root = (-halfB + sqrtD) / a;
        if (root < tMin || root > tMax) return 0;
    }
    rec->t = root;
    rec->p.x = r.orig.x + r.dir.x * root;
    rec->p.y = r.orig.y + r.dir.y * root;
    rec->p.z = r.orig.z + r.dir.z * root;
    double invRadius = 1.0 / s->radius;
    double nx = (rec->p.x - s->a.x) * invRadius;
    double ny = (rec->p.y - s->a.y) * invRadius;
    double nz = (rec->p.z - s->a.z) * invRadius;
    double dot = r.dir.x * nx + r.dir.y * ny + r.dir.z * nz;
    rec->frontFace = dot < 0;
    if (rec->frontFace) rec->normal = v(nx, ny, nz);
    else rec->normal = v(-nx, -ny, -nz);
    rec->mat = s->mat;
    return 1;
}

/* objects.go:98-133 */
static int plane_hit(const hittable *p, ray r, double tMin, double tMax, hitRecord *rec) {
    double denom = p->b.x * r.dir.x + p->b.y * r.dir.y + p->b.z * r.dir.z;
    if (fabs(denom) < 1e-6) return 0;
    double dx = p->a.x - r.orig.x, dy = p->a.y - r.orig.y, dz = p->a.z - r.orig.z;
    double t = (dx * p->b.x + dy * p->b.y + dz * p->b.z) / denom;
    if (t < tMin || t > tMax) return 0;
    rec->t = t;
    rec->p.x = r.orig.x + r.dir.x * t;
    rec->p.y = r.orig.y + r.dir.y * t;
    rec->p.z = r.orig.z + r.dir.z * t;
    rec->frontFace = denom < 0;
    if (rec->frontFace) rec->normal = p->b;
    else rec->normal = v(-p->b.x, -p->b.y, -p->b.z);
    rec->mat = p->mat;
    return 1;
}

/* objects.go:141-222 */
static int box_hit(const hittable *b, ray r, double tMin, double tMax, hitRecord *rec) {
    double t0 = tMin, t1 = tMax;
    for (int i = 0; i < 3; i++) {
        double invD, orig, minV, maxV;
        if (i == 0) { invD = 1 / r.dir.x; orig = r.orig.x; minV = b->a.x; maxV = b->b.x; }
        else if (i == 1) { invD = 1 / r.dir.y; orig = r.orig.y; minV = b->a.y; maxV = b->b.y; }
        else { invD = 1 / r.dir.z; orig = r.orig.z; minV = b->a.z; maxV = b->b.z; }
        double tNear = (minV - orig) * invD;
        double tFar = (maxV - orig) * invD;
        if (invD < 0) { double tmp = tNear; tNear = tFar; tFar = tmp; }
        if (tNear > t0) t0 = tNear;
        if (tFar < t1) t1 = tFar;
        if (t1 <= t0) return 0;
    }
    rec->t = t0;
    rec->p = vadd(r.orig, vmul(r.dir, t0)); /* ray.at, math.go:138-140 */
    double dxMin = rec->p.x - b->a.x, dxMax = b->b.x - rec->p.x;
    double dyMin = rec->p.y - b->a.y, dyMax = b->b.y - rec->p.y;
    double dzMin = rec->p.z - b->a.z, dzMax = b->b.z - rec->p.z;
    double minDist = dxMin;
    vec3 n = v(-1, 0, 0);
    if (dxMax < minDist) { minDist = dxMax; n = v(1, 0, 0); }
    if (dyMin < minDist) { minDist = dyMin; n = v(0, -1, 0); }
    if (dyMax < minDist) { minDist = dyMax; n = v(0, 1, 0); }
    if (dzMin < minDist) { minDist = dzMin; n = v(0, 0, -1); }
    if (dzMax < minDist) { n = v(0, 0, 1); }
    /* setFaceNormal, objects.go:17-24 */
    rec->frontFace = vdot(r.dir, n) < 0;
    if (rec->frontFace) rec->normal = n;
    else rec->normal = vmul(n, -1);
    rec->mat = b->mat;
    return 1;
}

static int obj_hit(const hittable *h, ray r, double tMin, double tMax, hitRecord *rec) {
    switch (h->kind) {
    case kSphere: return sphere_hit(h, r, tMin, tMax, rec);
    case kPlane: return plane_hit(h, r, tMin, tMax, rec);
    default: return box_hit(h, r, tMin, tMax, rec);
    }
}

/* objects.go:225-269 */
static int sceneToWorld(const ora_scene *sc, hittable *world) {
    int n = 0;
    for (int i = 0; i < sc->nobjects; i++) {
        const ora_object *o = &sc->objects[i];
        material mat;
        memset(&mat, 0, sizeof mat); /* missing id -> zero material = lambert black */
        if (o->material >= 0 && o->material < sc->nmaterials) mat = convertMaterial(&sc->materials[o->material]);
        vec3 pos = v(o->position[0], o->position[1], o->position[2]);
        vec3 size = v(o->size[0], o->size[1], o->size[2]);
        hittable h;
        memset(&h, 0, sizeof h);
        h.mat = mat;
        switch (o->type) {
        case 0: case 3:
            h.kind = kSphere; h.a = pos; h.radius = size.x;
            break;
        case 1:
            h.kind = kPlane; h.a = pos; h.b = v(0, 1, 0);
            break;
        case 2:
            h.kind = kBox; h.a = vsub(pos, vmul(size, 0.5)); h.b = vadd(pos, vmul(size, 0.5));
            break;
        default:
            continue;
        }
        world[n++] = h;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* camera.go                                                           */
/* ------------------------------------------------------------------ */

typedef struct {
    vec3 origin, lowerLeftCorner, horizontal, vertical, u, v, w;
    double lensRadius;
} camera;

/* camera.go:19-58 */
static camera newCamera(const ora_camera *c, int width, int height) {
    camera cam;
    double aspect = (double)width / (double)height;
    if (c->aspect_ratio != 0) aspect = c->aspect_ratio;
    double theta = c->fov * 3.141592653589793 / 180;
    double h = ora_tan(theta / 2);
    double viewportHeight = 2.0 * h;
    double viewportWidth = aspect * viewportHeight;
    vec3 origin = v(c->position[0], c->position[1], c->position[2]);
    vec3 target = v(c->target[0], c->target[1], c->target[2]);
    vec3 up = v(c->up[0], c->up[1], c->up[2]);
    vec3 w = vunit(vsub(origin, target));
    vec3 u = vunit(vcross(up, w));
    vec3 vVec = vcross(w, u);
    double focusDist = c->focus_dist;
    if (focusDist == 0) focusDist = vlength(vsub(origin, target));
    vec3 horizontal = vmul(u, viewportWidth * focusDist);
    vec3 vertical = vmul(vVec, viewportHeight * focusDist);
    vec3 llc = vsub(vsub(vsub(origin, vdiv(horizontal, 2)), vdiv(vertical, 2)), vmul(w, focusDist));
    cam.origin = origin; cam.lowerLeftCorner = llc; cam.horizontal = horizontal; cam.vertical = vertical;
    cam.u = u; cam.v = vVec; cam.w = w;
    cam.lensRadius = c->aperture / 2;
    return cam;
}

/* camera.go:60-74 */
static ray getRay(const camera *c, randSource *rng, double s, double t) {
    ray r;
    if (c->lensRadius > 0) {
        vec3 rd = vmul(randomInUnitSphere(rng), c->lensRadius);
        vec3 offset = vadd(vmul(c->u, rd.x), vmul(c->v, rd.y));
        r.orig = vadd(c->origin, offset);
        r.dir = vsub(vsub(vadd(vadd(c->lowerLeftCorner, vmul(c->horizontal, s)), vmul(c->vertical, t)), c->origin), offset);
        return r;
    }
    r.orig = c->origin;
    r.dir = vsub(vadd(vadd(c->lowerLeftCorner, vmul(c->horizontal, s)), vmul(c->vertical, t)), c->origin);
    return r;
}

/* ------------------------------------------------------------------ */
/* renderer.go                                                         */
/* ------------------------------------------------------------------ */

typedef struct {
    const hittable *world;
    int nworld;
    const ora_sky *sky;
    uint32_t segments, exit_scans;
} tracer;

/* renderer.go:56-92 */
static vec3 background(const ora_sky *sky, ray r) {
    if (sky->sky_type == 1) {
        vec3 horizon = v(sky->horizon[0], sky->horizon[1], sky->horizon[2]);
        vec3 zenith = v(sky->zenith[0], sky->zenith[1], sky->zenith[2]);
        double dirLen = sqrt(r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z);
        if (dirLen == 0) return horizon;
        double t = (r.dir.y / dirLen + 1.0) * 0.5;
        if (t < 0) t = 0;
        if (t > 1) t = 1;
        return v(horizon.x * (1 - t) + zenith.x * t, horizon.y * (1 - t) + zenith.y * t, horizon.z * (1 - t) + zenith.z * t);
    }
    if (sky->sky_type == 2) return v(sky->color[0], sky->color[1], sky->color[2]);
    return v(sky->background[0], sky->background[1], sky->background[2]);
}

/* renderer.go:286-404 */
static vec3 rayColorOpt(tracer *tr, ray r, int depth, randSource *rng, hitRecord *rec) {
    if (depth <= 0) return v(0, 0, 0);
    tr->segments++;

    const double tMin = 0.001;
    int hitAnything = 0;
    double closest = 1.79769313486231570814527423731704356798070e+308; /* math.MaxFloat64 */
    for (int i = 0; i < tr->nworld; i++) {
        if (obj_hit(&tr->world[i], r, tMin, closest, rec)) {
            hitAnything = 1;
            closest = rec->t;
        }
    }
    if (!hitAnything) return background(tr->sky, r);

    vec3 em = emitted(&rec->mat);
    vec3 attenuation;
    ray scattered;
    if (!scatter(&rec->mat, rng, r, rec, &attenuation, &scattered)) return em;

    if (rec->mat.typ == matDielectric) {
        if (rec->frontFace) {
            tr->exit_scans++;
            const double exitTMin = 0.0001;
            hitRecord exitRec;
            memset(&exitRec, 0, sizeof exitRec);
            int hitExit = 0;
            double exitT = 1.79769313486231570814527423731704356798070e+308;
            for (int i = 0; i < tr->nworld; i++) {
                hitRecord tempRec;
                memset(&tempRec, 0, sizeof tempRec);
                if (obj_hit(&tr->world[i], scattered, exitTMin, exitT, &tempRec)) {
                    if (tempRec.mat.typ == matDielectric && !tempRec.frontFace && tempRec.t < exitT) {
                        double dx = tempRec.p.x - rec->p.x;
                        double dy = tempRec.p.y - rec->p.y;
                        double dz = tempRec.p.z - rec->p.z;
                        double distSq = dx * dx + dy * dy + dz * dz;
                        if (distSq > 1e-8 && distSq < 1000.0) {
                            hitExit = 1;
                            exitT = tempRec.t;
                            exitRec = tempRec;
                        }
                    }
                }
            }
            if (hitExit) {
                double dx = exitRec.p.x - rec->p.x;
                double dy = exitRec.p.y - rec->p.y;
                double dz = exitRec.p.z - rec->p.z;
                double distance = sqrt(dx * dx + dy * dy + dz * dz);
                if (rec->mat.absorption.x > 0 || rec->mat.absorption.y > 0 || rec->mat.absorption.z > 0) {
                    attenuation.x = ora_exp(-rec->mat.absorption.x * distance);
                    attenuation.y = ora_exp(-rec->mat.absorption.y * distance);
                    attenuation.z = ora_exp(-rec->mat.absorption.z * distance);
                }
                scattered.orig = exitRec.p;
            }
        }
    }

    const int rrThreshold = 3;
    if (depth <= rrThreshold) {
        double maxAttenuation = ora_max(attenuation.x, ora_max(attenuation.y, attenuation.z));
        if (maxAttenuation < 1e-6) return em;
        double rrProb = ora_min(maxAttenuation, 0.95);
        if (Float64(rng) > rrProb) return em;
        attenuation.x /= rrProb;
        attenuation.y /= rrProb;
        attenuation.z /= rrProb;
    }

    hitRecord nextRec;
    memset(&nextRec, 0, sizeof nextRec);
    vec3 next = rayColorOpt(tr, scattered, depth - 1, rng, &nextRec);
    return v(em.x + attenuation.x * next.x, em.y + attenuation.y * next.y, em.z + attenuation.z * next.z);
}

/* renderer.go:190-221 + the amd64 behaviour of uint8(NaN) (-> 0) */
static uint8_t quantise(double val) {
    if (val < 0) val = 0;
    else if (val > 255.999) val = 255.999;
    if (is_nan(val)) return 0;
    return (uint8_t)val;
}

void ora_finish_pixel(const double sum[3], int32_t spp, uint8_t out[3]) {
    double invSamples = 1.0 / (double)spp;
    for (int c = 0; c < 3; c++) {
        double x = sum[c] * invSamples;
        x = sqrt(x);
        out[c] = quantise(x * 255.999);
    }
}

typedef struct {
    const ora_scene *sc;
    const ora_config *cfg;
    const hittable *world;
    int nworld;
    camera cam;
    double invWidth, invHeight, heightMinus1;
} frame;

static void frame_init(frame *f, const ora_scene *sc, const ora_config *cfg, hittable *world) {
    f->sc = sc; f->cfg = cfg;
    f->nworld = sceneToWorld(sc, world);
    f->world = world;
    f->cam = newCamera(&sc->camera, cfg->width, cfg->height);
    f->invWidth = 1.0 / (double)(cfg->width - 1);   /* renderer.go:95 */
    f->invHeight = 1.0 / (double)(cfg->height - 1); /* renderer.go:96 */
    f->heightMinus1 = (double)(cfg->height - 1);    /* renderer.go:98 */
}

/* one iteration of the sample loop, renderer.go:181-187 */
static vec3 sample_once(const frame *f, int x, int y, int s, uint32_t *nseg, uint32_t *nexit, uint32_t *ndraw) {
    randSource rng;
    rng.state = ora_stream_init(f->cfg->seed, (uint64_t)y * (uint64_t)f->cfg->width + (uint64_t)x, (uint64_t)s);
    rng.draws = 0;
    double flipY = f->heightMinus1 - (double)y;
    double u = ((double)x + Float64(&rng)) * f->invWidth;
    double vv = (flipY + Float64(&rng)) * f->invHeight;
    ray r = getRay(&f->cam, &rng, u, vv);
    tracer tr = {f->world, f->nworld, &f->sc->sky, 0, 0};
    hitRecord rec;
    memset(&rec, 0, sizeof rec);
    vec3 c = rayColorOpt(&tr, r, f->cfg->max_depth, &rng, &rec);
    *nseg += tr.segments;
    *nexit += tr.exit_scans;
    *ndraw += rng.draws;
    return c;
}

void ora_sample(const ora_scene *sc, const ora_config *cfg, int32_t x, int32_t y, int32_t s, double out_rgb[3],
                uint32_t *nseg, uint32_t *ndraw) {
    hittable *world = (hittable *)calloc((size_t)(sc->nobjects > 0 ? sc->nobjects : 1), sizeof(hittable));
    frame f;
    frame_init(&f, sc, cfg, world);
    uint32_t a = 0, b = 0, c = 0;
    vec3 col = sample_once(&f, x, y, s, &a, &b, &c);
    out_rgb[0] = col.x; out_rgb[1] = col.y; out_rgb[2] = col.z;
    if (nseg) *nseg = a;
    if (ndraw) *ndraw = c;
    free(world);
}

typedef struct {
    frame *f;
    int x0, y0, x1, y1;      /* window */
    int ntx, nty;            /* 32x32 tiles over the window's bounding tiles */
    int tx0, ty0;
    int next_tile;           /* guarded by mu */
    pthread_mutex_t mu;
    uint8_t *rgba; int32_t stride;
    double *accum; uint32_t *nseg; uint32_t *ndraw;
    uint64_t segments, exit_scans, draws;
} job;

static void *worker(void *arg) {
    job *jb = (job *)arg;
    frame *f = jb->f;
    const int tileSize = 32; /* renderer.go:132 */
    const int W = f->cfg->width, spp = f->cfg->spp;
    uint64_t segs = 0, exits = 0, draws = 0;
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        int t = jb->next_tile++;
        pthread_mutex_unlock(&jb->mu);
        if (t >= jb->ntx * jb->nty) break;
        int tx = (jb->tx0 + t % jb->ntx) * tileSize, ty = (jb->ty0 + t / jb->ntx) * tileSize;
        int xa = tx < jb->x0 ? jb->x0 : tx, ya = ty < jb->y0 ? jb->y0 : ty;
        int xb = tx + tileSize < jb->x1 ? tx + tileSize : jb->x1;
        int yb = ty + tileSize < jb->y1 ? ty + tileSize : jb->y1;
        for (int y = ya; y < yb; y++) {
            for (int x = xa; x < xb; x++) {
                vec3 col = v(0, 0, 0);
                uint32_t ps = 0, pe = 0, pd = 0;
                for (int s = 0; s < spp; s++) col = vadd(col, sample_once(f, x, y, s, &ps, &pe, &pd));
                segs += ps; exits += pe; draws += pd;
                size_t pi = (size_t)y * (size_t)W + (size_t)x;
                if (jb->accum) { jb->accum[3 * pi] = col.x; jb->accum[3 * pi + 1] = col.y; jb->accum[3 * pi + 2] = col.z; }
                if (jb->nseg) jb->nseg[pi] = ps;
                if (jb->ndraw) jb->ndraw[pi] = pd;
                if (jb->rgba) {
                    double sum[3] = {col.x, col.y, col.z};
                    uint8_t *px = jb->rgba + (size_t)y * (size_t)jb->stride + (size_t)x * 4;
                    ora_finish_pixel(sum, spp, px);
                    px[3] = 255;
                }
            }
        }
    }
    pthread_mutex_lock(&jb->mu);
    jb->segments += segs; jb->exit_scans += exits; jb->draws += draws;
    pthread_mutex_unlock(&jb->mu);
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int ora_render_window(const ora_scene *sc, const ora_config *cfg, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                      uint8_t *rgba, int32_t stride, double *accum, uint32_t *nseg, uint32_t *ndraw, ora_stats *stats) {
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > cfg->width) x1 = cfg->width;
    if (y1 > cfg->height) y1 = cfg->height;
    hittable *world = (hittable *)calloc((size_t)(sc->nobjects > 0 ? sc->nobjects : 1), sizeof(hittable));
    frame f;
    frame_init(&f, sc, cfg, world);

    /* renderer.go:106-112: the whole frame is cleared to opaque black first */
    if (rgba && x0 == 0 && y0 == 0 && x1 == cfg->width && y1 == cfg->height) {
        for (int y = 0; y < cfg->height; y++)
            for (int x = 0; x < cfg->width; x++) {
                uint8_t *px = rgba + (size_t)y * (size_t)stride + (size_t)x * 4;
                px[0] = px[1] = px[2] = 0; px[3] = 255;
            }
    }

    int workers = cfg->workers;
    if (workers <= 0) {
        workers = (int)sysconf(_SC_NPROCESSORS_ONLN);
        if (workers < 1) workers = 1;
        const char *env = getenv("PATHTRACER_WORKERS"); /* renderer.go:123-129 */
        if (env && *env) {
            int cw = atoi(env);
            if (cw > 0 && cw <= 128) workers = cw;
        }
    }

    job jb;
    memset(&jb, 0, sizeof jb);
    jb.f = &f;
    jb.x0 = x0; jb.y0 = y0; jb.x1 = x1; jb.y1 = y1;
    if (x1 > x0 && y1 > y0) {
        jb.tx0 = x0 / 32; jb.ty0 = y0 / 32;
        jb.ntx = (x1 + 31) / 32 - jb.tx0;
        jb.nty = (y1 + 31) / 32 - jb.ty0;
    }
    jb.rgba = rgba; jb.stride = stride; jb.accum = accum; jb.nseg = nseg; jb.ndraw = ndraw;
    pthread_mutex_init(&jb.mu, NULL);

    double t0 = now_s();
    pthread_t *th = (pthread_t *)calloc((size_t)workers, sizeof(pthread_t));
    for (int i = 0; i < workers; i++) pthread_create(&th[i], NULL, worker, &jb);
    for (int i = 0; i < workers; i++) pthread_join(th[i], NULL);
    double t1 = now_s();
    free(th);
    pthread_mutex_destroy(&jb.mu);
    free(world);

    if (stats) {
        int ww = x1 > x0 ? x1 - x0 : 0, hh = y1 > y0 ? y1 - y0 : 0;
        stats->samples = (uint64_t)ww * (uint64_t)hh * (uint64_t)(cfg->spp > 0 ? cfg->spp : 0);
        stats->segments = jb.segments;
        stats->exit_scans = jb.exit_scans;
        stats->draws = jb.draws;
        stats->seconds = t1 - t0;
        stats->workers = workers;
    }
    return 0;
}

int ora_render(const ora_scene *sc, const ora_config *cfg, uint8_t *rgba, int32_t stride, double *accum,
               uint32_t *nseg, uint32_t *ndraw, ora_stats *stats) {
    return ora_render_window(sc, cfg, 0, 0, cfg->width, cfg->height, rgba, stride, accum, nseg, ndraw, stats);
}

/* ------------------------------------------------------------------ */
/* unit-level exports                                                  */
/* ------------------------------------------------------------------ */

int ora_hit(int32_t kind, const double a[3], const double b[3], double radius, const double orig[3],
            const double dir[3], double tmin, double tmax, double out[8]) {
    hittable h;
    memset(&h, 0, sizeof h);
    h.kind = kind;
    h.a = v(a[0], a[1], a[2]);
    h.b = v(b[0], b[1], b[2]);
    h.radius = radius;
    ray r = {v(orig[0], orig[1], orig[2]), v(dir[0], dir[1], dir[2])};
    hitRecord rec;
    memset(&rec, 0, sizeof rec);
    int ok = obj_hit(&h, r, tmin, tmax, &rec);
    out[0] = rec.t;
    out[1] = rec.p.x; out[2] = rec.p.y; out[3] = rec.p.z;
    out[4] = rec.normal.x; out[5] = rec.normal.y; out[6] = rec.normal.z;
    out[7] = (double)rec.frontFace;
    return ok;
}

void ora_convert_material(const ora_material *m, double out[12]) {
    material r = convertMaterial(m);
    out[0] = (double)r.typ;
    out[1] = r.albedo.x; out[2] = r.albedo.y; out[3] = r.albedo.z;
    out[4] = r.rough; out[5] = r.ior;
    out[6] = r.emit.x; out[7] = r.emit.y; out[8] = r.emit.z;
    out[9] = r.absorption.x; out[10] = r.absorption.y; out[11] = r.absorption.z;
}

void ora_camera_setup(const ora_camera *c, int32_t width, int32_t height, double out[22]) {
    camera cam = newCamera(c, width, height);
    const vec3 *vs[7] = {&cam.origin, &cam.lowerLeftCorner, &cam.horizontal, &cam.vertical, &cam.u, &cam.v, &cam.w};
    for (int i = 0; i < 7; i++) { out[3 * i] = vs[i]->x; out[3 * i + 1] = vs[i]->y; out[3 * i + 2] = vs[i]->z; }
    out[21] = cam.lensRadius;
}

/* ------------------------------------------------------------------ */
/* post-process passes of the reference GPU backend (N4)               */
/* ------------------------------------------------------------------ */

/* gpu.go:22-47 */
float ora_aces_tonemap(float x) {
    if (x <= 0) return 0;
    const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
    double y = (double)x;
    double num = y * (a * y + b);
    double den = y * (c * y + d) + e;
    if (den <= 0) return 0;
    double r = num / den;
    if (r < 0) r = 0;
    else if (r > 1) r = 1;
    return (float)r;
}

static uint8_t to_u8_f32(float g) { /* uint8(g*255.0 + 0.5) in float32 arithmetic, gpu.go:2346-2348 */
    float v = g * 255.0f;
    v = v + 0.5f;
    return (uint8_t)v;
}

void ora_post_process(const ora_post_config *cfg, const double *accum, int32_t spp, uint8_t *rgba, int32_t stride,
                      int32_t w, int32_t h) {
    if (cfg->tonemap && accum) { /* gpu.go:2309-2350 */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                uint8_t *px = rgba + (size_t)y * (size_t)stride + (size_t)x * 4;
                for (int c = 0; c < 3; c++) {
                    float lin = (float)(accum[((size_t)y * (size_t)w + (size_t)x) * 3 + c] / (double)spp);
                    if (lin < 0) lin = 0;
                    float tm = ora_aces_tonemap(lin);
                    float g = (float)sqrt((double)tm);
                    if (g > 1) g = 1;
                    px[c] = to_u8_f32(g);
                }
                px[3] = 255;
            }
    }
    size_t bytes = (size_t)stride * (size_t)h;
    if (cfg->denoise && w > 2 && h > 2) { /* gpu.go:2355-2439 */
        uint8_t *sm = (uint8_t *)malloc(bytes);
        memcpy(sm, rgba, bytes);
        double twoSigmaS2 = 2 * cfg->sigma_s * cfg->sigma_s;
        double twoSigmaR2 = 2 * cfg->sigma_r * cfg->sigma_r;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const uint8_t *cp = rgba + (size_t)y * (size_t)stride + (size_t)x * 4;
                double cr = (double)cp[0] / 255.0, cg = (double)cp[1] / 255.0, cb = (double)cp[2] / 255.0;
                double sumR = 0, sumG = 0, sumB = 0, sumW = 0;
                for (int ky = -1; ky <= 1; ky++) {
                    int ny = y + ky;
                    if (ny < 0 || ny >= h) continue;
                    for (int kx = -1; kx <= 1; kx++) {
                        int nx = x + kx;
                        if (nx < 0 || nx >= w) continue;
                        const uint8_t *np = rgba + (size_t)ny * (size_t)stride + (size_t)nx * 4;
                        double nr = (double)np[0] / 255.0, ng = (double)np[1] / 255.0, nb = (double)np[2] / 255.0;
                        double ds2 = (double)(kx * kx + ky * ky);
                        double dr = cr - nr, dg = cg - ng, dbb = cb - nb;
                        double dr2 = dr * dr + dg * dg + dbb * dbb;
                        double ws = ora_exp(-ds2 / twoSigmaS2);
                        double wr = ora_exp(-dr2 / twoSigmaR2);
                        double wgt = ws * wr;
                        sumW += wgt;
                        sumR += nr * wgt;
                        sumG += ng * wgt;
                        sumB += nb * wgt;
                    }
                }
                uint8_t *op = sm + (size_t)y * (size_t)stride + (size_t)x * 4;
                if (sumW > 0) {
                    double v[3] = {sumR / sumW, sumG / sumW, sumB / sumW};
                    for (int c = 0; c < 3; c++) {
                        if (v[c] < 0) v[c] = 0;
                        else if (v[c] > 1) v[c] = 1;
                        op[c] = (uint8_t)(v[c] * 255.0 + 0.5);
                    }
                    op[3] = 255;
                } else {
                    memcpy(op, cp, 4);
                }
            }
        memcpy(rgba, sm, bytes);
        free(sm);
    }
    if (cfg->smooth && w > 2 && h > 2 && cfg->smooth_radius > 0 && cfg->smooth_strength > 0) { /* gpu.go:2444-2520 */
        int rad = cfg->smooth_radius;
        if (rad < 1) rad = 1;
        if (rad > 5) rad = 5;
        double str = cfg->smooth_strength;
        if (str < 0) str = 0;
        if (str > 1) str = 1;
        uint8_t *bl = (uint8_t *)calloc(bytes, 1); /* make([]byte, len(dst)): zero-filled, padding bytes included */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                double sumR = 0, sumG = 0, sumB = 0, count = 0;
                for (int ky = -rad; ky <= rad; ky++) {
                    int ny = y + ky;
                    if (ny < 0 || ny >= h) continue;
                    for (int kx = -rad; kx <= rad; kx++) {
                        int nx = x + kx;
                        if (nx < 0 || nx >= w) continue;
                        const uint8_t *np = rgba + (size_t)ny * (size_t)stride + (size_t)nx * 4;
                        sumR += (double)np[0];
                        sumG += (double)np[1];
                        sumB += (double)np[2];
                        count++;
                    }
                }
                if (count > 0) {
                    const uint8_t *cp = rgba + (size_t)y * (size_t)stride + (size_t)x * 4;
                    uint8_t *op = bl + (size_t)y * (size_t)stride + (size_t)x * 4;
                    double avg[3] = {sumR / count, sumG / count, sumB / count};
                    for (int c = 0; c < 3; c++) {
                        double out = (1 - str) * (double)cp[c] + str * avg[c];
                        if (out < 0) out = 0;
                        else if (out > 255) out = 255;
                        op[c] = (uint8_t)(out + 0.5);
                    }
                    op[3] = 255;
                }
            }
        /* copy(dst, blurred): the reference image has Stride == 4*w, so only pixel bytes exist */
        for (int y = 0; y < h; y++) memcpy(rgba + (size_t)y * (size_t)stride, bl + (size_t)y * (size_t)stride, (size_t)w * 4);
        free(bl);
    }
}
