// mcrt_device.h - device-side arithmetic of the path tracer (gfx950).
//
// Everything here is compiled with -ffp-contract=off: every float / double
// operation is a single correctly rounded IEEE operation in the order the
// reference performs it, so hit indices, hit distances and per-sample radiance
// can be compared bit for bit with the CPU.  Citations are to the reference
// checkout (source/...) and its vendored glm (include/glm/...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcrt_layout.h"

namespace mcrt {

#define MCRT_DEV __device__ __forceinline__

constexpr float kFltEpsilon = 1.1920928955078125e-07f;     // FLT_EPSILON
constexpr float kFltMax = 3.402823466e+38f;                 // FLOAT_INFINITY, source/core/math.h:32
constexpr float kMargin = kFltEpsilon * 20.0f;              // INTERSECTION_ERROR_MARGIN, source/core/math.h:34
constexpr float kTwoPiF = (float)6.28318530717958647692528676655900576;  // float(M_TWO_PI), source/core/math.h:24-26

// ------------------------------------------------------------------ vectors
struct V3 { float x, y, z; };
struct D3 { double x, y, z; };

MCRT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MCRT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
MCRT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
MCRT_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
MCRT_DEV V3 vdivs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// glm::dot(vec3): include/glm/detail/func_geometric.inl:48-55
MCRT_DEV float dot(V3 a, V3 b)
{
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// glm::cross: func_geometric.inl:68-79
MCRT_DEV V3 cross(V3 x, V3 y) { return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// glm::normalize = v * inversesqrt(dot(v,v)): func_geometric.inl:82-90, func_exponential.inl:44-50
MCRT_DEV V3 normalize(V3 v)
{
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
// glm::reflect: func_geometric.inl:104-110
MCRT_DEV V3 reflect(V3 I, V3 N) { return I - (N * dot(N, I)) * 2.0f; }

MCRT_DEV D3 dmk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
MCRT_DEV D3 operator+(D3 a, D3 b) { return dmk(a.x + b.x, a.y + b.y, a.z + b.z); }
MCRT_DEV D3 operator*(D3 a, D3 b) { return dmk(a.x * b.x, a.y * b.y, a.z * b.z); }
MCRT_DEV D3 operator*(D3 a, double s) { return dmk(a.x * s, a.y * s, a.z * s); }
MCRT_DEV D3 ddivs(D3 a, double s) { return dmk(a.x / s, a.y / s, a.z / s); }

// ------------------------------------------------------------- image access
// P is either an LDS pointer (address_space(3)) or a global pointer.
typedef const __attribute__((address_space(3))) uint32_t* LdsPtr;
typedef const uint32_t* GlbPtr;
// constant address space: loads with a wave-uniform address become scalar (SMEM) loads into SGPRs
typedef const __attribute__((address_space(4))) uint32_t* CstPtr;

// 16-byte loads (index must be a multiple of 4 words): ds_read_b128 / global_load_dwordx4 / s_load_dwordx4
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
MCRT_DEV U4 ld4(LdsPtr p, uint32_t i) { return *(const __attribute__((address_space(3))) U4*)(p + i); }
MCRT_DEV U4 ld4(GlbPtr p, uint32_t i) { return *(const U4*)(p + i); }
MCRT_DEV U4 ld4(CstPtr p, uint32_t i) { return *(const __attribute__((address_space(4))) U4*)(p + i); }
// The three 16-byte words of one 48-byte record (octree node or leaf record).  From global memory: ONE 64-bit address - index x 3,
// shifted by the add that forms the address - and the second and third word through the load's immediate offset; with the word index
// worked out in 32 bits (rec, rec + 4, rec + 8) the compiler built three addresses, each with its own 64-bit add, after a
// quarter-rate integer multiply.
static_assert((NODE_STRIDE == 12 || NODE_STRIDE == 16) && LEAF_STRIDE == 12, "ld_rec48 / ld_node read 48-byte records");
MCRT_DEV void ld_rec48(GlbPtr base, uint32_t idx, U4& a, U4& b, U4& c)
{
    const U4* p = (const U4*)base + (size_t)(idx * 3u);
    a = p[0]; b = p[1]; c = p[2];
}
// a node record: 48 bytes at a stride of NODE_STRIDE words
MCRT_DEV void ld_node(GlbPtr base, uint32_t idx, U4& a, U4& b, U4& c)
{
    if (NODE_STRIDE == 12) { ld_rec48(base, idx, a, b, c); return; }
    const U4* p = (const U4*)base + (size_t)idx * (NODE_STRIDE / 4);
    a = p[0]; b = p[1]; c = p[2];
}
MCRT_DEV void ld_rec48(LdsPtr base, uint32_t idx, U4& a, U4& b, U4& c)
{
    const uint32_t r = idx * 12u;
    a = ld4(base, r); b = ld4(base, r + 4); c = ld4(base, r + 8);
}
MCRT_DEV void ld_node(LdsPtr base, uint32_t idx, U4& a, U4& b, U4& c)
{
    const uint32_t r = idx * NODE_STRIDE;
    a = ld4(base, r); b = ld4(base, r + 4); c = ld4(base, r + 8);
}

struct TriRec { V3 v0, e1, e2; };
template <class P> MCRT_DEV TriRec load_tri(P big, uint32_t rec)
{
    const U4 a = ld4(big, rec), b = ld4(big, rec + 4), c = ld4(big, rec + 8);
    TriRec t;
    t.v0 = mk(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
    t.e1 = mk(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
    t.e2 = mk(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
    return t;
}

template <class P> MCRT_DEV float ldf(P p, uint32_t i) { return __uint_as_float(p[i]); }
template <class P> MCRT_DEV V3 ld3(P p, uint32_t i) { return mk(ldf(p, i), ldf(p, i + 1), ldf(p, i + 2)); }
template <class P> MCRT_DEV double ldd(P p, uint32_t i)
{
    uint64_t u = (uint64_t)p[i] | ((uint64_t)p[i + 1] << 32);
    return __longlong_as_double((long long)u);
}
template <class P> MCRT_DEV D3 ldd3(P p, uint32_t i) { return dmk(ldd(p, i), ldd(p, i + 2), ldd(p, i + 4)); }

// ---------------------------------------------------------------------- rng
// xorshift128+ and the 52-bit mantissa trick: source/core/randomization.h:23-43;
// float wrappers source/core/randomization.cpp:25-33.
struct Rng { uint64_t s0, s1; };

MCRT_DEV uint64_t rng_next(Rng& g)
{
    uint64_t x = g.s0;
    const uint64_t y = g.s1;
    g.s0 = y;
    x ^= x << 23;
    g.s1 = x ^ y ^ (x >> 17) ^ (y >> 26);
    return g.s1 + y;
}
MCRT_DEV double rng_double(Rng& g)
{
    uint64_t bits = (0x3FFull << 52) | (rng_next(g) >> 12);
    return __longlong_as_double((long long)bits) - 1.0;
}
MCRT_DEV float rng_float(Rng& g) { return (float)rng_double(g); }
MCRT_DEV float rng_float01(Rng& g) { return 0.0f + (1.0f - 0.0f) * (float)rng_double(g); }   // RandomFloat(0.0f, 1.0f)

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Counter (seed, pixel, sample) -> generator state; replaces std::random_device seeding
// (source/core/randomization.cpp:8-13).  mcrt_rng_seed exports the same mapping.
__host__ __device__ inline void sample_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t& s0, uint64_t& s1)
{
    uint64_t z = seed + (uint64_t)pixel * 0x9E3779B97F4A7C15ull + (uint64_t)sample * 0xD1B54A32D192ED03ull;
    s0 = mix64(z + 0x9E3779B97F4A7C15ull);
    s1 = mix64(z + 0x3C6EF372FE94F82Aull);
    if ((s0 | s1) == 0) s1 = 1;
}

// ------------------------------------------------------ float trig via double
// sinf/cosf/tanf/acosf with a fixed operation order (double minimax kernels,
// one rounding to float).  Within 1 ulp of glibc's, equal on > 99.9 % of
// inputs; the test oracle carries the same sequences, so device and host
// agree bit for bit.
MCRT_DEV double pm_ksin(double x)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double p = S5 + z * S6;
    p = S4 + z * p;
    p = S3 + z * p;
    p = S2 + z * p;
    p = S1 + z * p;
    return x + (x * z) * p;
}
MCRT_DEV double pm_kcos(double x)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double p = C5 + z * C6;
    p = C4 + z * p;
    p = C3 + z * p;
    p = C2 + z * p;
    p = C1 + z * p;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
MCRT_DEV void pm_sincos_d(float xf, double& s, double& c)
{
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, TWO_OVER_PI = 6.36619772367581382433e-01;
    double x = (double)xf;
    double kd = floor(x * TWO_OVER_PI + 0.5);
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    int q = (int)((long long)kd & 3);
    double sr = pm_ksin(r), cr = pm_kcos(r);
    if (q == 0) { s = sr; c = cr; }
    else if (q == 1) { s = cr; c = -sr; }
    else if (q == 2) { s = -sr; c = -cr; }
    else { s = -cr; c = sr; }
}
// (x - x) / (x - x) of the libm sources for an argument outside the domain: a quiet NaN.  Every caller has already
// established that x is NaN, infinite or out of range, so the division itself (a dozen instructions per site) is skipped;
// sign and payload of a NaN are not part of any comparison (non-finite samples are dropped, tests compare with isnan).
MCRT_DEV float pm_nan(float) { return __builtin_nanf(""); }
MCRT_DEV float pm_sinf(float x)
{
    if (!(fabsf(x) <= 8192.0f)) return pm_nan(x);
    double s, c; pm_sincos_d(x, s, c); return (float)s;
}
MCRT_DEV float pm_cosf(float x)
{
    if (!(fabsf(x) <= 8192.0f)) return pm_nan(x);
    double s, c; pm_sincos_d(x, s, c); return (float)c;
}
MCRT_DEV void pm_sincosf(float x, float& sf, float& cf)
{
    if (!(fabsf(x) <= 8192.0f)) { sf = cf = pm_nan(x); return; }
    double s, c; pm_sincos_d(x, s, c); sf = (float)s; cf = (float)c;
}
MCRT_DEV float pm_tanf(float x)
{
    if (!(fabsf(x) <= 8192.0f)) return pm_nan(x);
    double s, c; pm_sincos_d(x, s, c); return (float)(s / c);
}
MCRT_DEV double pm_asin_r(double z)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    double p = pS4 + z * pS5;
    p = pS3 + z * p;
    p = pS2 + z * p;
    p = pS1 + z * p;
    p = pS0 + z * p;
    p = z * p;
    double q = qS3 + z * qS4;
    q = qS2 + z * q;
    q = qS1 + z * q;
    q = 1.0 + z * q;
    return p / q;
}
MCRT_DEV float pm_acosf(float xf)
{
    // Three argument ranges (|x| < 0.5, x <= -0.5, x >= 0.5), each with the operation sequence the oracle's pm_acosf has,
    // but ONE evaluation of the rational R(z) and one square root shared by selects: a wavefront whose lanes fall into
    // different ranges executes the expensive part once instead of three times.
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, PI = 3.14159265358979311600e+00;
    const double x = (double)xf;
    const bool small = fabs(x) < 0.5, neg = x < 0.0;
    const double z = small ? x * x : ((neg ? 1.0 + x : 1.0 - x) * 0.5);
    const double base = small ? x : sqrt(z);
    const double a = base + base * pm_asin_r(z);
    const double r = small ? (PIO2_HI - a) + PIO2_LO : (neg ? PI - 2.0 * a : 2.0 * a);
    return (fabs(x) <= 1.0) ? (float)r : __builtin_nanf("");
}

// sinf(alpha) and tanf(beta) through ONE copy of the double-precision kernels (a two-trip loop the compiler must not unroll):
// same values as pm_sinf(alpha), pm_tanf(beta).
MCRT_DEV void pm_sin_and_tan(float alpha, float beta, float& sin_alpha, float& tan_beta)
{
    sin_alpha = tan_beta = 0.0f;
#pragma clang loop unroll(disable)
    for (int m = 0; m < 2; ++m) {
        const float x = m == 0 ? alpha : beta;
        float r = pm_nan(x);
        if (fabsf(x) <= 8192.0f) {
            double s, c;
            pm_sincos_d(x, s, c);
            r = (float)s;
            if (m == 1) r = (float)(s / c);
        }
        if (m == 0) sin_alpha = r; else tan_beta = r;
    }
}

// exp / log / pow in double for k_resolve (source/main.cpp:150-161), fixed operation order: the oracle's pm_exp / pm_log /
// pm_pow (oracle/mcrt_oracle.c) operation by operation, so the BGRA8 bytes of a resolve are identical on device and host.
// Within 1 ulp of libm (exp, log), ~1e-15 relative for pow = exp(y * log x).
MCRT_DEV uint64_t pm_bits(double x) { return (uint64_t)__double_as_longlong(x); }
MCRT_DEV double pm_from_bits(uint64_t u) { return __longlong_as_double((long long)u); }
MCRT_DEV double pm_log(double x)      // finite x > 0
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10,
                 LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01, LG3 = 2.857142874366239149e-01,
                 LG4 = 2.222219843214978396e-01, LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
                 LG7 = 1.479819860511658591e-01;
    long long k = 0;
    uint64_t u = pm_bits(x);
    if (u < 0x0010000000000000ull) { x = x * 18014398509481984.0; u = pm_bits(x); k = -54; }
    k += (long long)(u >> 52) - 1023;
    uint64_t m = u & 0x000FFFFFFFFFFFFFull;
    if (m >= 0x0006A09E667F3BCDull) { k += 1; m |= 0x3FE0000000000000ull; }
    else m |= 0x3FF0000000000000ull;
    const double f = pm_from_bits(m) - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    double t1 = LG4 + w * LG6;
    t1 = LG2 + w * t1;
    t1 = w * t1;
    double t2 = LG5 + w * LG7;
    t2 = LG3 + w * t2;
    t2 = LG1 + w * t2;
    t2 = z * t2;
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    const double dk = (double)k;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}
MCRT_DEV double pm_exp(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return pm_from_bits(0x7FF0000000000000ull);
    if (x < -7.45133219101941108420e+02) return 0.0;
    const long long k = (long long)(INV_LN2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double t = (double)k;
    const double hi = x - t * LN2_HI;
    const double lo = t * LN2_LO;
    const double r = hi - lo;
    const double tt = r * r;
    double p = P4 + tt * P5;
    p = P3 + tt * p;
    p = P2 + tt * p;
    p = P1 + tt * p;
    const double c = r - tt * p;
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021 && k <= 1022) return pm_from_bits(pm_bits(y) + ((uint64_t)k << 52));
    if (k > 1022) return pm_from_bits(pm_bits(y) + ((uint64_t)(k - 2) << 52)) * 4.0;
    return pm_from_bits(pm_bits(y) + ((uint64_t)(k + 1000) << 52)) * 9.33263618503218878990e-302;   // 2^-1000
}
MCRT_DEV double pm_pow(double x, double y)
{
    if (y == 1.0) return x;
    if (y == 0.0) return 1.0;
    if (x != x || y != y) return x + y;
    double sign = 1.0;
    if (x < 0.0) {
        if (y != floor(y)) return __builtin_nan("");
        if (fabs(y) < 9007199254740992.0 && floor(y * 0.5) * 2.0 != y) sign = -1.0;
        x = -x;
    }
    if (x == 0.0) return y > 0.0 ? 0.0 * sign : pm_from_bits(0x7FF0000000000000ull) * sign;
    if (x > 1.79769313486231570815e+308) return y > 0.0 ? x * sign : 0.0 * sign;
    return sign * pm_exp(y * pm_log(x));
}

// ------------------------------------------------------------- primitives
// Triangle::Intersects, source/core/triangle.h:25-62 (edges precomputed at upload).
MCRT_DEV bool tri_hit_rec(const TriRec& tr, V3 o, V3 d, float& t)
{
    const V3 v0 = tr.v0, edge1 = tr.e1, edge2 = tr.e2;
    V3 h = cross(d, edge2);
    float a = dot(edge1, h);
    if (fabsf(a) < kFltEpsilon) return false;
    float f = 1.0f / a;
    V3 s = o - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, edge1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot(edge2, q);
    return t > kFltEpsilon;
}
// The same operations without the early returns: in the octree walk every lane of the wavefront sits through the whole test
// anyway (some lane always gets to the end), and three nested branches cost more than the few instructions they skip.
// Same values: every quantity is computed by the same operation from the same operands; a rejected triangle just computes a
// t nobody reads (the reference writes it too, source/core/triangle.h:60).
MCRT_DEV bool tri_hit_flat(const TriRec& tr, V3 o, V3 d, float& t)
{
    const V3 h = cross(d, tr.e2);
    const float a = dot(tr.e1, h);
    const float f = 1.0f / a;
    const V3 s = o - tr.v0;
    const float u = f * dot(s, h);
    const V3 q = cross(s, tr.e1);
    const float v = f * dot(d, q);
    t = f * dot(tr.e2, q);
    return !(fabsf(a) < kFltEpsilon) & !(u < 0.0f || u > 1.0f) & !(v < 0.0f || u + v > 1.0f) & (t > kFltEpsilon);
}
template <class P> MCRT_DEV bool tri_hit(P big, uint32_t rec, V3 o, V3 d, float& t)
{
    return tri_hit_rec(load_tri(big, rec), o, d, t);
}

// SphereObject::Intersects, source/objects/sphere.h:16-52
template <class P> MCRT_DEV bool sphere_hit(P sm, uint32_t orec, V3 o, V3 d, float& t)
{
    const U4 rp = ld4(sm, orec + OBJ_RADIUS);        // radius, position.xyz
    float radius = __uint_as_float(rp.x);
    if (radius < kFltEpsilon) return false;
    float radiusSq = radius * radius;
    V3 L = mk(__uint_as_float(rp.y), __uint_as_float(rp.z), __uint_as_float(rp.w)) - o;
    float tca = dot(L, d);
    if (tca < 0) return false;
    float distanceSq = dot(L, L) - tca * tca;
    if (distanceSq > radiusSq) return false;
    float thc = sqrtf(radiusSq - distanceSq);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0) {
        t0 = t1;
        if (t0 < 0) return false;
    }
    t = t0;
    return true;
}

// AABB::IntersectsRay, source/core/aabb.h:49-96.  The reference widens the
// float quotients to double before min/max; widening preserves order, so the
// comparisons are done on the floats.
template <class P> MCRT_DEV bool gate_hit(P sm, uint32_t orec, V3 o, V3 d)
{
    float tmin = -kFltMax, tmax = kFltMax;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const U4 b0 = ld4(sm, orec + OBJ_AABB_MIN), b1 = ld4(sm, orec + OBJ_AABB_MIN + 4);
    const float mn[3] = { __uint_as_float(b0.x), __uint_as_float(b0.y), __uint_as_float(b0.z) };
    const float mx[3] = { __uint_as_float(b0.w), __uint_as_float(b1.x), __uint_as_float(b1.y) };
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0f) {
            float t1 = (mn[a] - oo[a]) / dd[a];
            float t2 = (mx[a] - oo[a]) / dd[a];
            float lo = (t2 < t1) ? t2 : t1;
            float hi = (t1 < t2) ? t2 : t1;
            tmin = (tmin < lo) ? lo : tmin;
            tmax = (hi < tmax) ? hi : tmax;
        }
    }
    return tmax >= tmin;
}

// Section profile (diagnostic builds only, -DMCRT_SECTION_PROFILE; tools/section_profile.py): per wavefront, the reference-clock
// time and the number of lanes with real work in each part of the loop.  Lane 0 of a wavefront keeps the sums in LDS (all the
// places that add are wave-uniform code) and adds them to g_prof when the kernel ends.
#ifdef MCRT_SECTION_PROFILE
enum : int { PF_ITER = 0, PF_PRE_T, PF_PRE_LANES, PF_CAST_T, PF_CAST_LANES, PF_SCAN_T, PF_SCAN_STEPS, PF_SCAN_LANES, PF_DEAL_T, PF_DEAL_STEPS, PF_DEAL_LANES,
              PF_POST_T, PF_POST_TRACE_LANES, PF_POST_SHADOW_LANES, PF_POST_HEAVY, PF_POST_HEAVY_LANES, PF_WAVES, PF_TOTAL_T, PF_IDLE_FETCH, PF_IDLE_DONE, PF_IDLE_BATCH,
              PF_COUNT };
__device__ unsigned long long g_prof[32];
MCRT_DEV unsigned long long* prof_slots()
{
    static __shared__ unsigned long long s[4 * 32];
    return s + (threadIdx.x >> 6) * 32;
}
#define PROF_NOW() __builtin_readcyclecounter()
#define PROF_ADD(slot_, value_) do { const unsigned long long pv_ = (unsigned long long)(value_); if ((threadIdx.x & 63u) == 0u) prof_slots()[slot_] += pv_; } while (0)
#else
#define PROF_ADD(slot_, value_) ((void)0)
#endif

struct Counters { uint32_t rays, nodes, tris, spheres, gates; };

struct Hit { uint32_t prim; uint32_t obj; uint32_t elem; float t; };   // obj == MISS on a miss

// (t, prim) beats the current best: strictly nearer, or equally near with a lower primitive id.
// A fresh `best` has t == FLT_MAX and prim == MISS and is only beaten by t < FLT_MAX, as in the
// reference's strict `<` against FLOAT_INFINITY.
MCRT_DEV bool nearer(float t, uint32_t prim, const Hit& best)
{
    return t < best.t || (t == best.t && best.prim != MISS && prim < best.prim);
}

// Scene::IntersectRay restated literally: source/scene.cpp:62-80 with
// TriangleMesh::Intersects, source/objects/mesh.cpp:9-41.
template <class PS, class PB, bool kCount>
MCRT_DEV Hit cast_brute(PS sm, PB big, const SceneView& v, V3 o, V3 d, Counters& cn)
{
    Hit best; best.prim = MISS; best.obj = MISS; best.elem = 0; best.t = kFltMax;
    if (kCount) cn.rays++;
    for (uint32_t i = 0; i < v.nobj; ++i) {
        uint32_t orec = v.off_obj + i * OBJ_STRIDE;
        uint32_t kind = sm[orec + OBJ_KIND];
        if (kind == MCRT_OBJECT_SPHERE) {
            float t;
            if (kCount) cn.spheres++;
            if (sphere_hit(sm, orec, o, d, t) && t < best.t) { best.obj = i; best.elem = 0; best.t = t; best.prim = sm[orec + OBJ_FIRST_PRIM]; }
        } else {
            if (kCount) cn.gates++;
            if (!gate_hit(sm, orec, o, d)) continue;
            uint32_t first = sm[orec + OBJ_FIRST_TRI], count = sm[orec + OBJ_TRI_COUNT];
            uint32_t elementIndex = 0;
            float nearest = kFltMax, hitDistance = kFltMax;
            for (uint32_t k = 0; k < count; ++k) {
                if (kCount) cn.tris++;
                if (tri_hit(big, v.off_tri + (first + k) * TRI_STRIDE, o, d, hitDistance) && hitDistance < nearest) {
                    nearest = hitDistance; elementIndex = k;
                }
            }
            if (nearest < kFltMax && nearest < best.t) { best.obj = i; best.elem = elementIndex; best.t = nearest; best.prim = sm[orec + OBJ_FIRST_PRIM] + elementIndex; }
        }
    }
    return best;
}

typedef __attribute__((address_space(3))) uint32_t* LdsRw;

MCRT_DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Nearest hit through the flat octree.  Same result as cast_brute: the
// candidate set is every primitive whose mesh gate passes, the winner is the
// lexicographic minimum of (t, primitive id) - "earlier object / earlier
// triangle wins ties", source/scene.cpp:73 and source/objects/mesh.cpp:22 -
// which does not depend on the order in which candidates are met.
// The mesh gate (source/objects/mesh.cpp:11-16) is a pure function of (ray,
// mesh); it is evaluated the first time the ray meets a triangle of that mesh
// and cached in two 64-bit masks.
//
// One iteration of run() is one node visit:
//   pop      the nearest stack entry not behind the best hit so far: (entry distance, node index);
//   fetch    the node's 48-byte record (three 16-byte loads issued together) and test its tight box;
//   descend  which of the node's octants does the ray pass, and where does it enter them?  That follows from the
//            node's own record - the stretch of the ray inside its tight box and the three mid planes of its CELL - so no
//            child record is touched (the first walk fetched and slab-tested all eight child boxes of every visited node:
//            256 bytes and sixteen loads a visit, 2.2x the algorithmic traffic): at most four octants, the one the ray is in
//            when it enters plus one per mid plane crossed, padded on the generous side (see visit()), pushed far to near;
//   leaves   ONE record of the list the lane is working off per step (the record of a list that has just begun is requested while
//            the octants are worked out): the other lanes of the wavefront wait for one record, never for the longest list among
//            64 nodes (two records per step measured 9 % slower, four 29 %).
// The walk is RESUMABLE: run() returns to the caller as soon as enough lanes of the wavefront have finished their
// ray, the caller shades those lanes, gives them their next ray and calls run() again; lanes in mid-traversal simply
// carry on.  Ray lengths in a big scene are close to exponentially distributed, so a wavefront that waits for its
// slowest lane on every query runs at a fraction of its width; refilling keeps it above the exit threshold.
struct ScratchStack {
    float t_[TRAVERSAL_STACK];
    uint32_t w_[TRAVERSAL_STACK];
    MCRT_DEV void put(int i, float t, uint32_t w) { t_[i] = t; w_[i] = w; }
    MCRT_DEV float t(int i) const { return t_[i]; }
    MCRT_DEV uint32_t get(int i) const { return w_[i]; }
    MCRT_DEV bool room(int, int) const { return false; }
    MCRT_DEV void put_fast(int i, float t, uint32_t w) { put(i, t, w); }
    MCRT_DEV void both(int i, float& t, uint32_t& w) const { t = t_[i]; w = w_[i]; }
};
// The first LDS_STACK_ENTRIES entries live in LDS (field-major, one dword column per thread of the 256-thread
// block: bank = thread id, conflict-free whatever the lanes' stack depths), the rest in scratch memory.  Scratch
// is dword-interleaved per wavefront, so 64 lanes pushing at 64 different depths touch 64 different cache lines:
// measured on the 1M-triangle scene, a scratch-only stack was most of the kernel's HBM traffic.
#ifndef MCRT_LDS_STACK
#define MCRT_LDS_STACK 16
#endif
#ifndef MCRT_WALK_BLOCKS
#define MCRT_WALK_BLOCKS 4
#endif
constexpr int LDS_STACK_ENTRIES = MCRT_LDS_STACK;     // 16 entries + the deal's arrays = the 36 KB per block that 18 entries took: four blocks per CU
constexpr int LDS_STACK_BLOCK = 256;
template <int kEntries> struct LdsStackT {
    __attribute__((address_space(3))) uint32_t* col;      // this thread's column
    float t_[TRAVERSAL_STACK - kEntries];
    uint32_t w_[TRAVERSAL_STACK - kEntries];
    MCRT_DEV void put(int i, float t, uint32_t w)
    {
        if (i < kEntries) { col[(i * 2 + 0) * LDS_STACK_BLOCK] = __float_as_uint(t); col[(i * 2 + 1) * LDS_STACK_BLOCK] = w; }
        else { t_[i - kEntries] = t; w_[i - kEntries] = w; }
    }
    MCRT_DEV float t(int i) const { return i < kEntries ? __uint_as_float(col[(i * 2 + 0) * LDS_STACK_BLOCK]) : t_[i - kEntries]; }
    MCRT_DEV uint32_t get(int i) const { return i < kEntries ? col[(i * 2 + 1) * LDS_STACK_BLOCK] : w_[i - kEntries]; }
    MCRT_DEV void both(int i, float& t, uint32_t& w) const
    {
        if (i < kEntries) { t = __uint_as_float(col[(i * 2 + 0) * LDS_STACK_BLOCK]); w = col[(i * 2 + 1) * LDS_STACK_BLOCK]; }
        else { t = t_[i - kEntries]; w = w_[i - kEntries]; }
    }
    // entries [i, i + n) all live in LDS: put_fast needs no check (one test per visit instead of one per push)
    MCRT_DEV bool room(int i, int n) const { return i + n <= kEntries; }
    MCRT_DEV void put_fast(int i, float t, uint32_t w) { col[(i * 2 + 0) * LDS_STACK_BLOCK] = __float_as_uint(t); col[(i * 2 + 1) * LDS_STACK_BLOCK] = w; }
};
typedef LdsStackT<LDS_STACK_ENTRIES> LdsStack;
// The same stack through a handle that is passed BY VALUE (the column pointer and the addresses of the deep part): a loop that works
// through it keeps the handle in registers, where a reference to the LdsStackT object - which lives in scratch memory because of its
// dynamically indexed arrays - made every access re-read the column pointer from scratch.
template <int kEntries> struct LdsStackRef {
    __attribute__((address_space(3))) uint32_t* col;
    float* t_;
    uint32_t* w_;
    MCRT_DEV explicit LdsStackRef(LdsStackT<kEntries>& s) : col(s.col), t_(s.t_), w_(s.w_) {}
    MCRT_DEV void put(int i, float t, uint32_t w) const
    {
        if (i < kEntries) { col[(i * 2 + 0) * LDS_STACK_BLOCK] = __float_as_uint(t); col[(i * 2 + 1) * LDS_STACK_BLOCK] = w; }
        else { t_[i - kEntries] = t; w_[i - kEntries] = w; }
    }
    MCRT_DEV void both(int i, float& t, uint32_t& w) const
    {
        if (i < kEntries) { t = __uint_as_float(col[(i * 2 + 0) * LDS_STACK_BLOCK]); w = col[(i * 2 + 1) * LDS_STACK_BLOCK]; }
        else {
            t = t_[i - kEntries]; w = w_[i - kEntries];
            // the deep part is scratch memory, i.e. VMEM: its loads are waited for HERE, inside the rare branch.  Left pending, every
            // later use of the popped entry carried a vmcnt(0) - which also waits for the leaf records requested just before
            asm volatile("" : "+v"(t), "+v"(w));
        }
    }
    MCRT_DEV bool room(int i, int n) const { return i + n <= kEntries; }
    MCRT_DEV void put_fast(int i, float t, uint32_t w) const { col[(i * 2 + 0) * LDS_STACK_BLOCK] = __float_as_uint(t); col[(i * 2 + 1) * LDS_STACK_BLOCK] = w; }
};
MCRT_DEV LdsStackRef<LDS_STACK_ENTRIES> stack_handle(LdsStack& s) { return LdsStackRef<LDS_STACK_ENTRIES>(s); }
MCRT_DEV ScratchStack& stack_handle(ScratchStack& s) { return s; }

// Reciprocal of a direction component for the slab and plane tests of the walk.  A zero (or tiny) component would give
// +-inf, and (face - origin) * inf is NaN when the origin lies exactly on the face - min/max drop the NaN and the node
// is culled although the ray runs inside it.  Clamped to +-2^100 every product of the walk stays FINITE (coordinates up
// to 1e8): 0 on the face, +-huge off it - the slab is open for a parallel ray between the faces and closed outside them,
// which is AABB::IntersectsRay's "skip the axis" (source/core/aabb.h:56) plus the containment the caller needs - and
// the padding keeps its meaning for a parallel ray: pad * 2^100 is finite and compares with (plane - origin) * 2^100.
constexpr float kWalkRcpMax = 1.2676506002282294e30f;     // 2^100
MCRT_DEV float slab_rcp(float d)
{
    const float r = __frcp_rn(d);
    return fabsf(r) <= kWalkRcpMax ? r : copysignf(kWalkRcpMax, d);
}

// Inclusive scans over the 64 lanes of the wavefront with DPP row shifts and row broadcasts (no LDS traffic).  All lanes active.
#define MCRT_DPP(X, CTRL, ROWS) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(X), (CTRL), (ROWS), 0xf, true))
MCRT_DEV uint32_t wave_scan_add(uint32_t x)
{
    x += MCRT_DPP(x, 0x111, 0xf);     // row_shr:1 (lanes shifted in from outside the row of 16 read 0)
    x += MCRT_DPP(x, 0x112, 0xf);     // row_shr:2
    x += MCRT_DPP(x, 0x114, 0xf);     // row_shr:4
    x += MCRT_DPP(x, 0x118, 0xf);     // row_shr:8
    x += MCRT_DPP(x, 0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    x += MCRT_DPP(x, 0x143, 0xc);     // row_bcast:31 into rows 2 and 3
    return x;
}
MCRT_DEV uint32_t wave_scan_max(uint32_t x)
{
    x = max(x, MCRT_DPP(x, 0x111, 0xf));
    x = max(x, MCRT_DPP(x, 0x112, 0xf));
    x = max(x, MCRT_DPP(x, 0x114, 0xf));
    x = max(x, MCRT_DPP(x, 0x118, 0xf));
    x = max(x, MCRT_DPP(x, 0x142, 0xa));
    x = max(x, MCRT_DPP(x, 0x143, 0xc));
    return x;
}
#undef MCRT_DPP

template <class PS, class PB, bool kCount> struct OctreeWalk {
    Hit best;
    uint64_t gate_known, gate_pass;
    int sp;               // entries on the stack
    uint32_t pleaf;       // next record of the leaf list the lane is working off, NO_REFS: none
    uint32_t pcnt;        // dealt walk only: records of that list still to be tested
    uint32_t epoch;       // dealt walk only: tag of the owner markers of the current deal (wave-uniform; lives as long as the LDS it tags)

    MCRT_DEV bool done() const { return sp == 0 && pleaf == NO_REFS; }      // the ray is finished, `best` is its answer
    MCRT_DEV void idle() { sp = 0; pleaf = NO_REFS; pcnt = 0; epoch = 0; gate_known = gate_pass = 0; best.prim = MISS; best.obj = MISS; best.elem = 0; best.t = kFltMax; }

    // Scenes with at most this many objects have every mesh gate evaluated when the ray starts - by all the lanes that start together -
    // instead of on first contact: evaluated lazily, about one lane of the wavefront meets a new mesh in every step, and the whole
    // wavefront sits through the six IEEE divisions of the gate for it (measured on the 5-object 1M-triangle scene: 15 % of a step).
    static constexpr uint32_t kEagerGates = 8;
    template <class St> MCRT_DEV void begin(PS sm, PB big, const SceneView& v, St& st, V3 o, V3 d, Counters& cn)
    {
        gate_known = 0; gate_pass = 0;
        // An origin far outside the scene (or NaN): an ulp of its coordinate is no longer small against the padding the walk decides
        // with, and the reference's own arithmetic decides inside that noise.  Such a ray is answered by the reference's loop over all
        // objects, right here; the lane comes out finished.  (SceneView::far_limit; rays that start inside the scene - every bounce
        // and shadow ray, any camera within ~26 scene sizes - never take this branch.)
        if (!(fabsf(o.x) <= v.far_limit && fabsf(o.y) <= v.far_limit && fabsf(o.z) <= v.far_limit)) {
            best = cast_brute<PS, PB, kCount>(sm, big, v, o, d, cn);
            sp = 0; pleaf = NO_REFS; pcnt = 0;
            return;
        }
        best.prim = MISS; best.obj = MISS; best.elem = 0; best.t = kFltMax;
        if (kCount) cn.rays++;
        if (v.nobj <= kEagerGates) {
            for (uint32_t i = 0; i < v.nobj; ++i) {
                const uint32_t orec = v.off_obj + i * OBJ_STRIDE;
                if (sm[orec + OBJ_KIND] == (uint32_t)MCRT_OBJECT_SPHERE) continue;
                if (kCount) cn.gates++;
                if (gate_hit(sm, orec, o, d)) gate_pass |= 1ull << i;
            }
            gate_known = ~0ull;
        }
        pleaf = NO_REFS; pcnt = 0;
        st.put(0, 0.0f, 0u); sp = 1;          // the root
    }

    // The octants of a node that the ray passes inside [tb0, tb1], pushed far to near (cw: the node's child word, rc: its cell words).
    template <class St> static MCRT_DEV void push_octants(St& st, int& sp, float best_t, uint32_t cw, const U4& rc, float tb0, float tb1, V3 o, V3 inv, V3 padt, V3 dts, uint32_t flip)
    {
        const uint32_t mask = cw >> 24;
        {
            const uint32_t base = cw & 0x00FFFFFFu;
            const float scale = __uint_as_float(rc.w & NODE_SCALE_MASK);
            // Which octants, in which order?  A straight line meets each of the cell's three mid planes once, so inside [tb0, tb1] it
            // passes at most four octants: the one it is in at tb0 and one more per mid plane crossed (Revelles' parametric walk).
            // In ray order (bit a of r set = beyond the mid plane of axis a): r starts with the planes already behind at tb0; every
            // plane crossed before tb1 flips its bit, nearest crossing first.  Everything is decided with the padding on the
            // generous side: a plane counts as behind only when it is more than the padding behind, as crossed when it comes within
            // the padding of [tb0, tb1], and the octant beyond it is entered the padding early.  A hit point always lies in the padded
            // cell of the octant that is current at its distance (per axis it is on the same side of the mid plane or within the
            // padding of it), and the builder references a primitive from every child whose padded cell it overlaps - so the
            // primitive is met there, and the entry distance is not behind it (ties included).  "The padding" of these decisions
            // (padt) is HALF the builder's: the distances below carry a rounding error of a few ulps of max(|origin|, |cell|), and
            // with the builder's full padding a slow ray that runs 2e-7 OUTSIDE the padded neighbour cell was routed through that
            // neighbour - which does not reference the primitive - for the rest of its way (round 2's hole; tests: pad_margin_rays).
            // With half of it the walk's "within reach of the child" is always inside the builder's.  The octant test this replaces
            // worked out all eight padded boxes (max3 / min3 / compare each) and pushed in a loop: ~225 instructions, now ~90.
            const float cmn[3] = { __uint_as_float(rc.x), __uint_as_float(rc.y), __uint_as_float(rc.z) };
            const float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z }, pp[3] = { padt.x, padt.y, padt.z }, dd[3] = { dts.x, dts.y, dts.z };
            uint32_t r0 = 0;
            float c[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // the mid plane from the cell's corner: the cell is the root cell scaled by 2^-depth, half of dts * scale away
                const float tm = __fmaf_rn(dd[k] * scale, 0.5f, (cmn[k] - oo[k]) * ii[k]);
                const float enter = tm - pp[k];                       // where the half beyond the plane begins for this ray
                const bool behind = enter < tb0 - 2.0f * pp[k] ? true : false;      // tm < tb0 - pad
                r0 |= behind ? (1u << k) : 0u;
                c[k] = (!behind && enter <= tb1) ? enter : __builtin_inff();
            }
            // the three crossings in order (a three-element sorting network; the axis bit travels with its distance)
            float k0 = c[0], k1 = c[1], k2 = c[2];
            uint32_t b0 = 1u, b1 = 2u, b2 = 4u;
            // (compare-and-select, no branches: the compiler had turned two of the three swaps into exec-mask regions)
#define MCRT_SORT2(KA, KB, BA, BB)                                                               \
            {                                                                                    \
                const bool sw = (KB) < (KA);                                                     \
                const float lo = sw ? (KB) : (KA), hi = sw ? (KA) : (KB);                        \
                const uint32_t blo = sw ? (BB) : (BA), bhi = sw ? (BA) : (BB);                   \
                (KA) = lo; (KB) = hi; (BA) = blo; (BB) = bhi;                                    \
            }
            MCRT_SORT2(k0, k1, b0, b1)
            MCRT_SORT2(k1, k2, b1, b2)
            MCRT_SORT2(k0, k1, b0, b1)
#undef MCRT_SORT2
            const uint32_t r1 = r0 ^ b0, r2 = r1 ^ b1, r3 = r2 ^ b2;
            // far to near; an octant whose entry lies behind the best hit so far (or that is not entered at all: +inf) is left out
            if (st.room(sp, 4)) {
                // Room for all four in the LDS part of the stack (the rule, deeper stacks are rare): every candidate is WRITTEN at the
                // current top, and the top moves up only for a valid one - an invalid candidate is overwritten by the next write or stays
                // above the top, where nobody looks.  Four unconditional writes and four adds instead of four nested exec-mask regions.
#define MCRT_PUSH_OCTANT(R, T, OK)                                                               \
                {                                                                                \
                    const uint32_t oct = (R) ^ flip;                                             \
                    const uint32_t child = base + __popc(mask & ((1u << oct) - 1u));             \
                    st.put_fast(sp, (T), child);                                                 \
                    sp += ((OK) && ((mask >> oct) & 1u)) ? 1 : 0;                                \
                }
                MCRT_PUSH_OCTANT(r3, fmaxf(tb0, k2), k2 <= best_t)
                MCRT_PUSH_OCTANT(r2, fmaxf(tb0, k1), k1 <= best_t)
                MCRT_PUSH_OCTANT(r1, fmaxf(tb0, k0), k0 <= best_t)
                MCRT_PUSH_OCTANT(r0, tb0, true)
#undef MCRT_PUSH_OCTANT
            } else {
#define MCRT_PUSH_OCTANT(R, T)                                                                   \
                {                                                                                \
                    const uint32_t oct = (R) ^ flip;                                             \
                    if ((mask >> oct) & 1u) {                                                    \
                        const uint32_t child = base + __popc(mask & ((1u << oct) - 1u));         \
                        st.put(sp, (T), child);                                                  \
                        ++sp;                                                                    \
                    }                                                                            \
                }
                if (k2 <= best_t) MCRT_PUSH_OCTANT(r3, fmaxf(tb0, k2))
                if (k1 <= best_t) MCRT_PUSH_OCTANT(r2, fmaxf(tb0, k1))
                if (k0 <= best_t) MCRT_PUSH_OCTANT(r1, fmaxf(tb0, k0))
                MCRT_PUSH_OCTANT(r0, tb0)
#undef MCRT_PUSH_OCTANT
            }
        }
    }

    // One step of the walk, the same two pieces of code for every lane of the wavefront:
    //   node step   (lanes that are not in the middle of a leaf list) pop, fetch the node's record, test its tight box, push the
    //               octants the ray passes; the node's own list, if any, becomes the lane's leaf list;
    //   leaf step   (lanes with a leaf list, the ones that just got theirs included) ONE record: fetch, test, advance.
    // A lane with a long list stays in leaf steps and does not pop meanwhile, so per ray the order of events is that of a
    // sequential walk; the wavefront, though, never sits through the longest list among its 64 nodes: a step costs one node
    // step plus one record whatever the lists look like (the chunked list loop it replaces ran at 13-17 of 64 lanes).
    // inv: clamped reciprocals of the direction; padt = pad * |inv| per axis; dts = root cell size * inv per axis; flip: bit a set where the ray runs
    // towards smaller coordinates on axis a (sign of inv, so that +-0 components are treated consistently).
    template <class St> MCRT_DEV void visit(PS sm, PB nodes, PB leaves, const SceneView& v, St& st, V3 o, V3 d, V3 inv, V3 padt, V3 dts, uint32_t flip,
                        Counters& cn)
    {
        U4 a, b, c;
        bool loaded = false;
        if (pleaf != NO_REFS) {
            // in the middle of a list: request the record now, the node steps of the other lanes run while it travels
            const uint32_t r = pleaf * LEAF_STRIDE;
            a = ld4(leaves, r); b = ld4(leaves, r + 4); c = ld4(leaves, r + 8);
            loaded = true;
        } else {
            // next entry that was not entered behind the best hit so far
            float t_in;
            uint32_t w_in;
            do {
                if (sp == 0) return;
                --sp;
                st.both(sp, t_in, w_in);
            } while (t_in > best.t);
            const uint32_t rec = w_in * NODE_STRIDE;
            const U4 ra = ld4(nodes, rec), rb = ld4(nodes, rec + 4), rc = ld4(nodes, rec + 8);
            // all three loads in one round trip: without this the compiler sinks the third into the `mask != 0` branch, behind the tight test
            asm volatile("" ::"v"(rc.x), "v"(rc.y), "v"(rc.z), "v"(rc.w));
            if (kCount) cn.nodes++;
            // the node's own tight box against [0, best.t]: [tb0, tb1] is the stretch of the ray inside it
            float tb0 = 0.0f, tb1 = best.t;
            {
                float p = (__uint_as_float(ra.x) - o.x) * inv.x, q = (__uint_as_float(ra.w) - o.x) * inv.x;
                tb0 = fmaxf(tb0, fminf(p, q)); tb1 = fminf(tb1, fmaxf(p, q));
                p = (__uint_as_float(ra.y) - o.y) * inv.y; q = (__uint_as_float(rb.x) - o.y) * inv.y;
                tb0 = fmaxf(tb0, fminf(p, q)); tb1 = fminf(tb1, fmaxf(p, q));
                p = (__uint_as_float(ra.z) - o.z) * inv.z; q = (__uint_as_float(rb.y) - o.z) * inv.z;
                tb0 = fmaxf(tb0, fminf(p, q)); tb1 = fminf(tb1, fmaxf(p, q));
                if (!(tb0 <= tb1)) return;
            }
            pleaf = rb.w;
            if (pleaf != NO_REFS) {
                // the node's own list: its first record travels while the octants are worked out
                const uint32_t r = pleaf * LEAF_STRIDE;
                a = ld4(leaves, r); b = ld4(leaves, r + 4); c = ld4(leaves, r + 8);
                asm volatile("" ::: "memory");
                loaded = true;
            }
            if (rb.z >> 24) push_octants(st, sp, best.t, rb.z, rc, tb0, tb1, o, inv, padt, dts, flip);
        }

        if (pleaf != NO_REFS) {
            if (!loaded) {
                const uint32_t r = pleaf * LEAF_STRIDE;
                a = ld4(leaves, r); b = ld4(leaves, r + 4); c = ld4(leaves, r + 8);
            }
            const uint32_t prim = c.y & ~REF_LAST;
            const uint32_t obj = c.z & ~LEAF_SPHERE;
            const uint32_t orec = v.off_obj + obj * OBJ_STRIDE;
            float t;
            bool h;
            if (c.z & LEAF_SPHERE) {
                if (kCount) cn.spheres++;
                h = sphere_hit(sm, orec, o, d, t);
            } else {
                // mesh gate, evaluated once per (ray, mesh) on first contact (objects beyond the 64 the cache holds: on every contact)
                const uint64_t bit = obj < MAX_OBJECTS_GATED ? 1ull << obj : 0ull;
                bool pass;
                if (obj >= MAX_OBJECTS_GATED) {
                    if (kCount) cn.gates++;
                    pass = gate_hit(sm, orec, o, d);
                } else {
                    if (!(gate_known & bit)) {
                        if (kCount) cn.gates++;
                        gate_known |= bit;
                        if (gate_hit(sm, orec, o, d)) gate_pass |= bit;
                    }
                    pass = (gate_pass & bit) != 0ull;
                }
                h = false;
                if (pass) {
                    if (kCount) cn.tris++;
                    TriRec tr;
                    tr.v0 = mk(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
                    tr.e1 = mk(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
                    tr.e2 = mk(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
                    h = tri_hit_flat(tr, o, d, t);
                }
            }
            if (h && nearer(t, prim, best)) {
                best.prim = prim; best.obj = obj; best.elem = prim - sm[orec + OBJ_FIRST_PRIM]; best.t = t;
            }
            pleaf = (c.y & REF_LAST) ? NO_REFS : pleaf + 1u;
        }
    }

    // per-ray constants of the walk
    static MCRT_DEV void ray_setup(const SceneView& v, V3 d, V3& inv, V3& padt, V3& dts, uint32_t& flip)
    {
        // Node and octant culling is conservative (boxes and cells are padded); reciprocals are clamped, see slab_rcp.
        inv = mk(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
        padt = mk(v.walk_pad * fabsf(inv.x), v.walk_pad * fabsf(inv.y), v.walk_pad * fabsf(inv.z));
        dts = mk(v.root_size[0] * inv.x, v.root_size[1] * inv.y, v.root_size[2] * inv.z);
        flip = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    }

    // Advance every unfinished lane.  refill == 0: until all lanes of the wavefront are finished; otherwise
    // return once fewer than refill/64 of the lanes that came in unfinished are still walking.
    template <class St> MCRT_DEV void run(PS sm, PB big, const SceneView& v, St& st, V3 o, V3 d, Counters& cn, uint32_t refill)
    {
        V3 inv, padt, dts;
        uint32_t flip;
        ray_setup(v, d, inv, padt, dts, flip);
        const PB nodes = big + v.off_node;
        const PB leaves = big + v.off_leaf;
        const uint32_t entered = (uint32_t)__popcll(__ballot(!done()));
        const uint32_t keep = (entered * refill + 63u) / 64u;    // leave when fewer than this many are walking
        for (;;) {
            const uint32_t walking = (uint32_t)__popcll(__ballot(!done()));
            if (walking == 0 || walking < keep) break;
            if (!done()) visit(sm, nodes, leaves, v, st, o, d, inv, padt, dts, flip, cn);
        }
    }

    // ---------------------------------------------------------------- the walk with DEALT leaf records
    // The step above leaves the leaf half of every step to the lanes that hold a list: about a third of the wavefront, and a list of
    // three records keeps its lane out of the node steps for three rounds.  Here the records that the lanes of a wavefront have
    // pending - (ray, record) pairs - are laid end to end and dealt to ALL 64 lanes, one pair each: lane j finds the lane that owns
    // pair j (a prefix sum over the list lengths, which ride in the node records), fetches that lane's ray through ds_bpermute and
    // the record from HBM, tests it, and merges a hit into the owner's slot with one 64-bit LDS atomic min on (t, primitive id) -
    // nearest first, then the lower id: `nearer`, i.e. the reference's tie-break (source/scene.cpp:73, source/objects/mesh.cpp:22).
    // A list is consumed in the step in which its node was visited (unless the wavefront has more than 64 records pending), so every
    // walking lane takes part in every node step, and the leaf tests run at the width of the wavefront.  The candidate set is the
    // same or larger (a lane culls with the best distance as of the previous step), the winner is the minimum of the same keys.
    struct Deal {
        __attribute__((address_space(3))) unsigned long long* key;   // [block] per thread: (ordered t bits << 32) | primitive id of the ray's best hit
        LdsRw slot;                                                  // [block] 64 per wavefront: owner markers of the deal
    };
    // float bits -> unsigned integers in the same order (a sphere can report t = -0.0, source/objects/sphere.h:44-49; every other hit is positive)
    static MCRT_DEV uint32_t t_ord(float t) { const uint32_t b = __float_as_uint(t); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
    static MCRT_DEV float ord_t(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
    // the object that owns primitive `prim`: the last one whose first primitive id is not beyond it (ids are assigned in object order)
    static MCRT_DEV uint32_t obj_of_prim(PS sm, const SceneView& v, uint32_t prim)
    {
        uint32_t lo = 0, hi = v.nobj - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (sm[v.off_obj + mid * OBJ_STRIDE + OBJ_FIRST_PRIM] <= prim) lo = mid; else hi = mid - 1u;
        }
        return lo;
    }

    template <class St> MCRT_DEV void begin_dealt(PS sm, PB big, const SceneView& v, St& st, const Deal& L, V3 o, V3 d, Counters& cn)
    {
        begin(sm, big, v, st, o, d, cn);
        // the answer lives in LDS from here on (a far origin has it already: the loop over all objects ran in begin)
        L.key[threadIdx.x] = ((unsigned long long)t_ord(best.t) << 32) | (unsigned long long)best.prim;
    }
    // the finished ray's answer back into `best`
    MCRT_DEV void collect_dealt(PS sm, const SceneView& v, const Deal& L)
    {
        const unsigned long long k = L.key[threadIdx.x];
        best.prim = (uint32_t)k;
        if (best.prim == MISS) { best.obj = MISS; best.elem = 0; best.t = kFltMax; }
        else {
            best.obj = obj_of_prim(sm, v, best.prim);
            best.t = ord_t((uint32_t)(k >> 32));
            best.elem = best.prim - sm[v.off_obj + best.obj * OBJ_STRIDE + OBJ_FIRST_PRIM];
        }
    }

    // The loop is software-pipelined: the record of the node a lane visits NEXT is requested before the dealt records of this step are
    // tested (records first, node second: loads return in order and are consumed in that order), so each of the two HBM round trips
    // of a step runs under the arithmetic of the other half.  A node requested but not yet looked at when run_dealt hands control
    // back goes back on the stack (its entry is still there: a pop only moves the stack pointer).
    template <class StObj> MCRT_DEV void run_dealt(PS sm, PB big, const SceneView& v, StObj& st_obj, const Deal& L, V3 o, V3 d, Counters& cn, uint32_t refill)
    {
        // The loop works on LOCAL copies of the walk's variables and on a by-value handle of the stack: the OctreeWalk and stack objects
        // live in scratch memory (the stack's deep part is indexed dynamically), and a loop that went through them re-read the stack
        // pointer, the list counter and the column pointer from scratch in every step - each reload waiting for ALL loads in flight,
        // the node and leaf records included.
        auto st = stack_handle(st_obj);
        int sp = this->sp;
        uint32_t pleaf = this->pleaf, pcnt = this->pcnt, epoch = this->epoch;
        float best_t = this->best.t;
        const uint32_t gate_lo = (uint32_t)gate_pass;
        V3 inv, padt, dts;
        uint32_t flip;
        ray_setup(v, d, inv, padt, dts, flip);
        const PB nodes = big + v.off_node;
        const PB leaves = big + v.off_leaf;
        const uint32_t lane = threadIdx.x & 63u, wbase = threadIdx.x & ~63u;
        const bool eager = v.nobj <= kEagerGates;            // every gate of the ray is known (begin): one dword travels with the ray
        const uint32_t entered = (uint32_t)__popcll(__ballot(sp != 0 || pleaf != NO_REFS));
        const uint32_t keep = (entered * refill + 63u) / 64u;
        bool have = false;                                   // ra, rb, rc hold the record of the node this lane visits next
        U4 ra, rb, rc;
        ra.x = ra.y = ra.z = ra.w = 0u; rb = ra; rc = ra;
        // pop the next entry that was not entered behind the best hit so far and request its record
#define MCRT_REQUEST_NODE()                                                                       \
        {                                                                                         \
            uint32_t w_in = 0u;                                                                   \
            if (pleaf == NO_REFS && sp > 0) {                                                     \
                float t_in;                                                                       \
                bool found = false;                                                               \
                do {                                                                              \
                    --sp;                                                                         \
                    st.both(sp, t_in, w_in);                                                      \
                    found = !(t_in > best_t);                                                     \
                } while (!found && sp > 0);                                                       \
                have = found;                                                                     \
            }                                                                                     \
            /* every lane loads (a lane without a node: the root's record, a cache hit): three loads that are issued on every path can be */ \
            /* counted, and the wait for the leaf records becomes vmcnt(3) instead of vmcnt(0) - which would wait for these loads too */ \
            ld_node(nodes, have ? w_in : 0u, ra, rb, rc);                                          \
            asm volatile("" ::: "memory");                                                        \
        }
        MCRT_REQUEST_NODE()
        for (;;) {
            const uint32_t walking = (uint32_t)__popcll(__ballot(have || sp != 0 || pleaf != NO_REFS));
            if (walking == 0 || walking < keep) break;

            // ---- node step, first half: the tight box of the node whose record has arrived
            bool descend = false;
            uint32_t cw = 0;
            U4 cell = rc;
            float tb0 = 0.0f, tb1 = 0.0f;
            const bool any_node = __ballot(have) != 0ull;
#ifdef MCRT_SECTION_PROFILE
            PROF_ADD(PF_SCAN_STEPS, 1); PROF_ADD(PF_SCAN_LANES, __popcll(__ballot(have))); PROF_ADD(PF_IDLE_DONE, 64 - walking);
#endif
            // the requested record is waited for HERE, on every path: left pending where no lane looks at a node, it made every later write to
            // one of its registers - the pop loop, the next request - wait for all loads in flight, the leaf records included
            asm volatile("" ::"v"(ra.x), "v"(rb.x), "v"(rc.x));
            if (have) {
                have = false;
                if (kCount) cn.nodes++;
                tb0 = 0.0f; tb1 = best_t;
                float p = (__uint_as_float(ra.x) - o.x) * inv.x, q = (__uint_as_float(ra.w) - o.x) * inv.x;
                tb0 = fmaxf(tb0, fminf(p, q)); tb1 = fminf(tb1, fmaxf(p, q));
                p = (__uint_as_float(ra.y) - o.y) * inv.y; q = (__uint_as_float(rb.x) - o.y) * inv.y;
                tb0 = fmaxf(tb0, fminf(p, q)); tb1 = fminf(tb1, fmaxf(p, q));
                p = (__uint_as_float(ra.z) - o.z) * inv.z; q = (__uint_as_float(rb.y) - o.z) * inv.z;
                tb0 = fmaxf(tb0, fminf(p, q)); tb1 = fminf(tb1, fmaxf(p, q));
                if (tb0 <= tb1) {
                    pleaf = rb.w;
                    pcnt = rc.w & NODE_COUNT_MASK;
                    cw = rb.z;
                    descend = (cw >> 24) != 0u;
                }
            }

            // ---- the deal: pair j of the wavefront's pending records goes to lane j
            const uint32_t cnt = pleaf != NO_REFS ? pcnt : 0u;
            const uint32_t incl = wave_scan_add(cnt), excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#ifndef MCRT_DEAL_MIN
#define MCRT_DEAL_MIN 1
#endif
            // wave-uniform: the records are dealt when there are enough of them to fill most of the wavefront, or when no lane has a node to look at
            const bool dealing = total >= (uint32_t)MCRT_DEAL_MIN || (total != 0u && !any_node);
            bool mine = false;
            uint32_t own = 0;
            V3 po = o, pd = d;
            uint32_t pgate = 0;
            U4 a, b, c;
            a.x = a.y = a.z = a.w = 0u; b = a; c = a;
            if (dealing) {
#ifdef MCRT_SECTION_PROFILE
                PROF_ADD(PF_DEAL_STEPS, 1); PROF_ADD(PF_DEAL_LANES, total < 64u ? total : 64u);
                PROF_ADD(PF_POST_HEAVY_LANES, __popcll(__ballot(descend)));
#endif
                // the markers of this deal carry a fresh tag: what earlier deals left in the slots reads as "none" (no zeroing pass)
                epoch += 64u;
                if (epoch == 0u) { L.slot[threadIdx.x] = 0u; wave_sync(); epoch = 64u; }     // the tag wrapped (2^26 deals): start over
                if (cnt != 0u && excl < 64u) L.slot[wbase + excl] = epoch | lane;
                wave_sync();
                const uint32_t m = L.slot[threadIdx.x];
                // owners ascend with the pairs: the marker at a list's first pair fills forward (pair 0 always carries one)
                own = wave_scan_max((m & ~63u) == epoch ? (m & 63u) : 0u);
                mine = lane < total;
                const int src = (int)(own << 2);
                const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(pleaf - excl)) + lane;     // the owner's list, (lane - excl[owner]) records in
                po = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o.y))),
                        __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o.z))));
                pd = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.y))),
                        __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.z))));
                if (eager) pgate = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)gate_lo);
                {
                    // (lanes without a pair load record 0: loads issued on every path can be counted by the compiler's wait insertion)
                    ld_rec48(leaves, mine ? r : 0u, a, b, c);
                    asm volatile("" ::: "memory");             // the records travel while the octants are worked out
                }
                // the owners move on: what is dealt now is as good as tested
                if (cnt != 0u) {
                    const uint32_t room = excl < 64u ? 64u - excl : 0u;
                    const uint32_t used = cnt < room ? cnt : room;
                    pleaf += used; pcnt -= used;
                    if (pcnt == 0u) pleaf = NO_REFS;
                }
            }

            // ---- node step, second half: the octants the ray passes, far to near; then the next node is requested
            if (descend) push_octants(st, sp, best_t, cw, cell, tb0, tb1, o, inv, padt, dts, flip);
            MCRT_REQUEST_NODE()

            // ---- the dealt records
            if (dealing) {
                if (mine) {
                    // every word of the three loads stays claimed up to here: the record's spare word (c.w) had been handed out as a
                    // temporary right after the loads were issued, and writing it meant waiting for them - before the octants
                    asm volatile("" ::"v"(a.x), "v"(b.x), "v"(c.w));
                    bool h = false;
                    float t = 0.0f;
                    const uint32_t prim = c.y & ~REF_LAST;
                    const uint32_t obj = c.z & ~LEAF_SPHERE;
                    const uint32_t orec = v.off_obj + __umul24(obj, OBJ_STRIDE);       // obj < MAX_OBJECTS: the full-rate multiply
                    if (c.z & LEAF_SPHERE) {
                        if (kCount) cn.spheres++;
                        h = sphere_hit(sm, orec, po, pd, t);
                    } else {
                        bool pass;
                        if (eager) pass = ((pgate >> obj) & 1u) != 0u;
                        else {
                            // the gate (source/objects/mesh.cpp:11-16) is a pure function of ray and mesh: evaluated on the contact
                            if (kCount) cn.gates++;
                            pass = gate_hit(sm, orec, po, pd);
                        }
                        if (pass) {
                            if (kCount) cn.tris++;
                            TriRec tr;
                            tr.v0 = mk(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
                            tr.e1 = mk(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
                            tr.e2 = mk(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
                            h = tri_hit_flat(tr, po, pd, t);
                        }
                    }
                    h = h && t < kFltMax;                          // a fresh best is beaten by t < FLT_MAX only (`nearer`)
                    if (h) atomicMin((unsigned long long*)(L.key + (wbase + own)), ((unsigned long long)t_ord(t) << 32) | (unsigned long long)prim);
                }
                wave_sync();
                best_t = ord_t((uint32_t)(L.key[threadIdx.x] >> 32));     // what the next pops and slabs cull with
            }
        }
#undef MCRT_REQUEST_NODE
        if (have) ++sp;          // requested, not looked at: the entry is still on the stack
        this->sp = sp; this->pleaf = pleaf; this->pcnt = pcnt; this->epoch = epoch; this->best.t = best_t;
    }
};

template <class PS, class PB, bool kCount>
MCRT_DEV Hit cast_octree(PS sm, PB big, const SceneView& v, V3 o, V3 d, Counters& cn)
{
    OctreeWalk<PS, PB, kCount> w;
    ScratchStack st;
    w.begin(sm, big, v, st, o, d, cn);
    w.run(sm, big, v, st, o, d, cn, 0u);
    return w.best;
}

// ---------------------------------------------------------------- wave-uniform scan
// Nearest hit for scenes small enough that a whole wavefront is better off walking the object
// list in lock step than letting 64 lanes chase 64 different tree paths: every lane visits the
// same object / triangle at the same time, so the geometry arrives through SCALAR loads (SGPR
// operands, no LDS or VGPR traffic, no divergent control flow), a mesh is skipped for the whole
// wave when no lane passes its gate, and the loop order IS the reference's order (objects, then
// triangles), which gives its first-wins tie-break for free (source/scene.cpp:71-76).
//
// Arithmetic: rejections may be decided on cheap approximations (reciprocal-multiply instead of
// IEEE division) as long as they are conservative; anything that can become the reported hit -
// and every borderline case - is recomputed with the reference's exact operation sequence.

// Two triangles at a time, component-interleaved (PAIR_* layout): every float2 operation below is
// one packed instruction whose second operand is an SGPR pair straight out of the scalar load.
typedef float F2 __attribute__((ext_vector_type(2)));
typedef uint32_t U8 __attribute__((ext_vector_type(8)));
typedef uint32_t U16 __attribute__((ext_vector_type(16)));

MCRT_DEV F2 f2(uint32_t a, uint32_t b) { F2 r; r.x = __uint_as_float(a); r.y = __uint_as_float(b); return r; }
MCRT_DEV F2 dot2(F2 ax, F2 ay, F2 az, F2 bx, F2 by, F2 bz)
{
    F2 tx = ax * bx, ty = ay * by, tz = az * bz;
    return tx + ty + tz;
}

// The exact tail of Triangle::Intersects (source/core/triangle.h:44-61) from its exactly computed
// intermediates a = dot(edge1,h), x = dot(s,h), y = dot(d,q), z = dot(edge2,q).
MCRT_DEV bool tri_exact_tail(float a, float x, float y, float z, float best_t, float& t_out)
{
    const float f = 1.0f / a;
    const float u = f * x;
    if (u < 0.0f || u > 1.0f) return false;
    const float w = f * y;
    if (w < 0.0f || u + w > 1.0f) return false;
    const float t = f * z;
    if (!(t > kFltEpsilon) || !(t < best_t)) return false;
    t_out = t;
    return true;
}

// "can this triangle beat best_t?" decided on a reciprocal-multiply estimate (relative error of
// u', v', t' below 4e-7) whenever the answer is a clear no.
MCRT_DEV bool tri_maybe(float a, float x, float y, float z, float best_t)
{
    const float fa = __builtin_amdgcn_rcpf(a);
    const float ua = fa * x, va = fa * y, ta = fa * z;
    const float m = 4.0e-6f;
    return !(fabsf(a) < kFltEpsilon) && !(ua < -m) && !(ua > 1.0f + m) && !(va < -m) && !(ua + va > 1.0f + 2.0f * m) &&
           (ta > 0.5f * kFltEpsilon) && (ta < best_t * (1.0f + m));
}

// AABB::IntersectsRay (source/core/aabb.h:49-96): decided on reciprocal-multiply slabs when the
// outcome is clear, on the exact quotients otherwise.  `inv` = approximate 1/d per axis; `exact_only`
// is set for rays with a zero or denormal-range direction component (an axis the reference skips, or approximate
// slabs that overflow).
MCRT_DEV bool gate_hit_fast(CstPtr rec, V3 o, V3 d, V3 inv, bool exact_only)
{
    const U8 b = *(const __attribute__((address_space(4))) U8*)(rec + OBJ_AABB_MIN);
    const float mn[3] = { __uint_as_float(b.s0), __uint_as_float(b.s1), __uint_as_float(b.s2) };
    const float mx[3] = { __uint_as_float(b.s3), __uint_as_float(b.s4), __uint_as_float(b.s5) };
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z }, ii[3] = { inv.x, inv.y, inv.z };
    float tmin = -kFltMax, tmax = kFltMax;
    // every direction component is a normal, non-zero number here (exact_only covers the rest), so the products are
    // finite or +-inf, never NaN, and plain min / max instructions do
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t1 = (mn[a] - oo[a]) * ii[a];
        const float t2 = (mx[a] - oo[a]) * ii[a];
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    const float diff = tmax - tmin;
    const float tol = 4.0e-6f * (fabsf(tmax) + fabsf(tmin)) + 1.0e-30f;
    const bool clear_pass = diff > tol, clear_fail = diff < -tol;
    if (!exact_only && (clear_pass || clear_fail)) return clear_pass;
    // exact: float quotients, as the reference computes them
    tmin = -kFltMax; tmax = kFltMax;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0f) {
            const float t1 = (mn[a] - oo[a]) / dd[a];
            const float t2 = (mx[a] - oo[a]) / dd[a];
            const float lo = (t2 < t1) ? t2 : t1;
            const float hi = (t1 < t2) ? t2 : t1;
            tmin = (tmin < lo) ? lo : tmin;
            tmax = (hi < tmax) ? hi : tmax;
        }
    }
    return tmax >= tmin;
}

// Exact tails of a triangle pair.  A ray is a candidate for at most one triangle of a pair almost always (the two
// halves of a quad), so the candidate's intermediates are selected first and ONE exact tail - the IEEE division and
// the reference's comparisons - serves both; a lane that is a candidate for both (it grazes the shared edge) takes
// the second tail afterwards.  Order and strictness are those of the sequential loop: triangle k, then k + 1.
MCRT_DEV void pair_exact(F2 a, F2 x, F2 y, F2 z, uint32_t obj, uint32_t k, Hit& best)
{
    const bool cx = tri_maybe(a.x, x.x, y.x, z.x, best.t), cy = tri_maybe(a.y, x.y, y.y, z.y, best.t);
    if (cx || cy) {
        float t;
        const bool second = !cx;
        if (tri_exact_tail(second ? a.y : a.x, second ? x.y : x.x, second ? y.y : y.x, second ? z.y : z.x, best.t, t)) {
            best.obj = obj; best.elem = k + (second ? 1u : 0u); best.t = t;
        }
        if (cx && cy && tri_maybe(a.y, x.y, y.y, z.y, best.t) && tri_exact_tail(a.y, x.y, y.y, z.y, best.t, t)) { best.obj = obj; best.elem = k + 1u; best.t = t; }
    }
}

template <bool kCount>
MCRT_DEV Hit cast_scan(CstPtr sm, CstPtr big, const SceneView& v, V3 o, V3 d, Counters& cn, bool live = true)
{
    Hit best; best.prim = MISS; best.obj = MISS; best.elem = 0; best.t = kFltMax;
    if (kCount && live) cn.rays++;
    const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const float tiny = 1.0e-20f;
    const bool exact_only = fabsf(d.x) < tiny || fabsf(d.y) < tiny || fabsf(d.z) < tiny;
    CstPtr rec = sm + v.off_obj;
    for (uint32_t i = 0; i < v.nobj; ++i, rec += OBJ_STRIDE) {
        const U8 head = *(const __attribute__((address_space(4))) U8*)rec;   // kind, first_tri, tri_count, first_prim, radius, pos
        if (head.s0 == MCRT_OBJECT_SPHERE) {
            // SphereObject::Intersects, source/objects/sphere.h:16-52
            if (kCount && live) cn.spheres++;
            const float radius = __uint_as_float(head.s4);
            if (live && !(radius < kFltEpsilon)) {
                const float radiusSq = radius * radius;
                const V3 L = mk(__uint_as_float(head.s5), __uint_as_float(head.s6), __uint_as_float(head.s7)) - o;
                const float tca = dot(L, d);
                const float distanceSq = dot(L, L) - tca * tca;
                if (!(tca < 0) && !(distanceSq > radiusSq)) {
                    const float thc = sqrtf(radiusSq - distanceSq);
                    float t0 = tca - thc, t1 = tca + thc;
                    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
                    if (t0 < 0) t0 = t1;
                    if (!(t0 < 0) && t0 < best.t) { best.obj = i; best.elem = 0; best.t = t0; }
                }
            }
        } else {
            if (kCount && live) cn.gates++;
            const bool pass = live && gate_hit_fast(rec, o, d, inv, exact_only);
            if (__ballot(pass) == 0ull) continue;             // nobody in the wave needs this mesh
            const uint32_t count = head.s2;
            CstPtr pr = big + v.off_pair + (size_t)rec[OBJ_FIRST_PAIR] * PAIR_STRIDE;
            if (pass) {
                for (uint32_t k = 0; k < count; k += 2, pr += PAIR_STRIDE) {
                    const U16 c0 = *(const __attribute__((address_space(4))) U16*)pr;
                    const U4 c1 = *(const __attribute__((address_space(4))) U4*)(pr + 16);
                    const F2 v0x = f2(c0.s0, c0.s1), v0y = f2(c0.s2, c0.s3), v0z = f2(c0.s4, c0.s5);
                    const F2 e1x = f2(c0.s6, c0.s7), e1y = f2(c0.s8, c0.s9), e1z = f2(c0.sa, c0.sb);
                    const F2 e2x = f2(c0.sc, c0.sd), e2y = f2(c0.se, c0.sf), e2z = f2(c1.x, c1.y);
                    // Triangle::Intersects, source/core/triangle.h:31-60, both triangles at once
                    const F2 hx = d.y * e2z - e2y * d.z, hy = d.z * e2x - e2z * d.x, hz = d.x * e2y - e2x * d.y;   // cross(d, edge2)
                    const F2 a = dot2(e1x, e1y, e1z, hx, hy, hz);
                    const F2 sx = o.x - v0x, sy = o.y - v0y, sz = o.z - v0z;
                    const F2 x = dot2(sx, sy, sz, hx, hy, hz);
                    const F2 qx = sy * e1z - e1y * sz, qy = sz * e1x - e1z * sx, qz = sx * e1y - e1x * sy;          // cross(s, edge1)
                    const F2 y = dot2(F2(d.x), F2(d.y), F2(d.z), qx, qy, qz);
                    const F2 z = dot2(e2x, e2y, e2z, qx, qy, qz);
                    if (kCount) cn.tris += (k + 1 < count) ? 2u : 1u;
                    pair_exact(a, x, y, z, i, k, best);
                }
            }
        }
    }
    if (best.obj != MISS) best.prim = sm[v.off_obj + best.obj * OBJ_STRIDE + OBJ_FIRST_PRIM] + best.elem;
    return best;
}

// ---------------------------------------------------------------- wave-cooperative flat scan
// Same candidate set and same answers as cast_scan, different division of labour.  In cast_scan every
// lane of the wavefront sits through every triangle of every mesh that ANY lane's ray may touch; with 64
// incoherent rays that is all of them, while a single ray passes the gates of a fifth of the triangles
// (measured: 31 % active lanes).  Here the (ray, triangle) pairs that actually need testing - for each
// mesh, the rays that passed its gate times its triangles - are laid end to end and dealt out evenly
// to the lanes: lane r takes pairs r, r + n, r + 2n ... (n = active lanes), fetches "its" ray from a
// per-wavefront LDS table, tests one triangle and, on a hit, merges (t, object, element) into the ray's
// slot with one 64-bit LDS atomic min - the key order IS the reference's tie-break (nearest, then earlier
// object, then earlier triangle: source/scene.cpp:73, source/objects/mesh.cpp:22).
constexpr uint32_t FLAT_AREA_WORDS = 64u * 12u;     // per wavefront: 64 rays x 8 words, 64 keys x 2 words, 64 list entries x 1 word (+ pad)
constexpr uint32_t FLAT_ELEM_BITS = 20;
#ifndef MCRT_FLAT_MIN
#define MCRT_FLAT_MIN 24
#endif
constexpr uint32_t FLAT_MIN_SCAN_LANES = MCRT_FLAT_MIN;
template <bool kCount>
MCRT_DEV Hit cast_flat(CstPtr sm, LdsPtr big, LdsRw area, const SceneView& v, V3 o, V3 d, Counters& cn, bool live = true)
{
    if (kCount && live) cn.rays++;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t act = __ballot(1);
    const uint32_t nact = (uint32_t)__popcll(act);
    const uint64_t below = (1ull << lane) - 1ull;
    const uint32_t rank = (uint32_t)__popcll(act & below);
    LdsRw rays = area;                       // [64][8]
    LdsRw keys = area + 64u * 8u;            // [64] u64
    LdsRw list = area + 64u * 10u;           // [64]
    {
        U4 a, b;
        a.x = __float_as_uint(o.x); a.y = __float_as_uint(o.y); a.z = __float_as_uint(o.z); a.w = 0u;
        b.x = __float_as_uint(d.x); b.y = __float_as_uint(d.y); b.z = __float_as_uint(d.z); b.w = 0u;
        *(__attribute__((address_space(3))) U4*)(rays + lane * 8u) = a;
        *(__attribute__((address_space(3))) U4*)(rays + lane * 8u + 4u) = b;
        *(__attribute__((address_space(3))) unsigned long long*)(keys + lane * 2u) = 0x7F7FFFFFFFFFFFFFull;   // (FLT_MAX, MISS)
    }
    Hit best; best.prim = MISS; best.obj = MISS; best.elem = 0; best.t = kFltMax;     // spheres and widely needed meshes: tested by the ray's own lane
    const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const float tiny = 1.0e-20f;
    const bool exact_only = fabsf(d.x) < tiny || fabsf(d.y) < tiny || fabsf(d.z) < tiny;
    CstPtr rec = sm + v.off_obj;
    for (uint32_t i = 0; i < v.nobj; ++i, rec += OBJ_STRIDE) {
        const U8 head = *(const __attribute__((address_space(4))) U8*)rec;   // kind, first_tri, tri_count, first_prim, radius, pos
        if (head.s0 == MCRT_OBJECT_SPHERE) {
            // SphereObject::Intersects, source/objects/sphere.h:16-52
            if (kCount && live) cn.spheres++;
            const float radius = __uint_as_float(head.s4);
            if (live && !(radius < kFltEpsilon)) {
                const float radiusSq = radius * radius;
                const V3 L = mk(__uint_as_float(head.s5), __uint_as_float(head.s6), __uint_as_float(head.s7)) - o;
                const float tca = dot(L, d);
                const float distanceSq = dot(L, L) - tca * tca;
                if (!(tca < 0) && !(distanceSq > radiusSq)) {
                    const float thc = sqrtf(radiusSq - distanceSq);
                    float t0 = tca - thc, t1 = tca + thc;
                    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
                    if (t0 < 0) t0 = t1;
                    if (!(t0 < 0) && t0 < best.t) { best.obj = i; best.elem = 0; best.t = t0; }
                }
            }
            continue;
        }
        if (kCount && live) cn.gates++;
        const bool pass = live && gate_hit_fast(rec, o, d, inv, exact_only);
        const uint64_t mask = __ballot(pass);
        if (mask == 0ull) continue;                                   // nobody in the wave needs this mesh
        const uint32_t count = head.s2, first = head.s1;
        if (count == 0u) continue;
        if ((uint32_t)__popcll(mask) >= FLAT_MIN_SCAN_LANES) {
            // Most of the wavefront needs this mesh (the walls of a closed room): the lock-step pair loop of
            // cast_scan costs 38 instructions per lane and triangle against ~100 for a dealt pair, so it wins
            // from about 24 passing lanes up.
#ifdef MCRT_SECTION_PROFILE
            const unsigned long long pt0 = PROF_NOW();
            PROF_ADD(PF_SCAN_STEPS, (count + 1u) / 2u); PROF_ADD(PF_SCAN_LANES, (unsigned long long)__popcll(mask) * ((count + 1u) / 2u));
#endif
            if (pass) {
                CstPtr pr = (CstPtr)v.big_img + v.off_pair + (size_t)rec[OBJ_FIRST_PAIR] * PAIR_STRIDE;
                for (uint32_t k = 0; k < count; k += 2, pr += PAIR_STRIDE) {
                    const U16 c0 = *(const __attribute__((address_space(4))) U16*)pr;
                    const U4 c1 = *(const __attribute__((address_space(4))) U4*)(pr + 16);
                    const F2 v0x = f2(c0.s0, c0.s1), v0y = f2(c0.s2, c0.s3), v0z = f2(c0.s4, c0.s5);
                    const F2 e1x = f2(c0.s6, c0.s7), e1y = f2(c0.s8, c0.s9), e1z = f2(c0.sa, c0.sb);
                    const F2 e2x = f2(c0.sc, c0.sd), e2y = f2(c0.se, c0.sf), e2z = f2(c1.x, c1.y);
                    const F2 hx = d.y * e2z - e2y * d.z, hy = d.z * e2x - e2z * d.x, hz = d.x * e2y - e2x * d.y;   // cross(d, edge2)
                    const F2 a = dot2(e1x, e1y, e1z, hx, hy, hz);
                    const F2 sx = o.x - v0x, sy = o.y - v0y, sz = o.z - v0z;
                    const F2 x = dot2(sx, sy, sz, hx, hy, hz);
                    const F2 qx = sy * e1z - e1y * sz, qy = sz * e1x - e1z * sx, qz = sx * e1y - e1x * sy;          // cross(s, edge1)
                    const F2 y = dot2(F2(d.x), F2(d.y), F2(d.z), qx, qy, qz);
                    const F2 z = dot2(e2x, e2y, e2z, qx, qy, qz);
                    if (kCount) cn.tris += (k + 1 < count) ? 2u : 1u;
                    pair_exact(a, x, y, z, i, k, best);
                }
            }
#ifdef MCRT_SECTION_PROFILE
            PROF_ADD(PF_SCAN_T, PROF_NOW() - pt0);
#endif
            continue;
        }
#ifdef MCRT_SECTION_PROFILE
        const unsigned long long pt1 = PROF_NOW();
#endif
        if (pass) list[(uint32_t)__popcll(mask & below)] = lane;       // compact the passing rays
        wave_sync();
        const uint32_t total = (uint32_t)__popcll(mask) * count;
        const float inv_count = 1.0f / (float)count;
        for (uint32_t w = rank; w < total; w += nact) {
            // slot = w / count, k = w % count  (count is wave-uniform, w < 2^24: one float multiply and a fix-up)
            uint32_t slot = (uint32_t)((float)w * inv_count);
            if (slot * count > w) --slot;
            else if ((slot + 1u) * count <= w) ++slot;
            const uint32_t k = w - slot * count;
            const uint32_t src = list[slot];
            const U4 ra = *(const __attribute__((address_space(3))) U4*)(rays + src * 8u);
            const U4 rb = *(const __attribute__((address_space(3))) U4*)(rays + src * 8u + 4u);
            const V3 ro = mk(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z));
            const V3 rd = mk(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z));
            const TriRec tr = load_tri(big, v.off_tri + (first + k) * TRI_STRIDE);
            if (kCount) cn.tris++;
            // Triangle::Intersects, source/core/triangle.h:31-60
            const V3 h = cross(rd, tr.e2);
            const float a = dot(tr.e1, h);
            const V3 sv = ro - tr.v0;
            const float x = dot(sv, h);
            const V3 q = cross(sv, tr.e1);
            const float y = dot(rd, q);
            const float z = dot(tr.e2, q);
            float t;
            if (tri_maybe(a, x, y, z, kFltMax) && tri_exact_tail(a, x, y, z, kFltMax, t)) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((i << FLAT_ELEM_BITS) | k);
                atomicMin((unsigned long long*)(keys + src * 2u), key);
            }
        }
        wave_sync();                                                  // the list is reused by the next mesh
#ifdef MCRT_SECTION_PROFILE
        PROF_ADD(PF_DEAL_T, PROF_NOW() - pt1); PROF_ADD(PF_DEAL_STEPS, (total + nact - 1u) / nact); PROF_ADD(PF_DEAL_LANES, total);
#endif
    }
    const unsigned long long key = *(__attribute__((address_space(3))) unsigned long long*)(keys + lane * 2u);
    const float tm = __uint_as_float((uint32_t)(key >> 32));
    const uint32_t om = ((uint32_t)key) >> FLAT_ELEM_BITS, em = ((uint32_t)key) & ((1u << FLAT_ELEM_BITS) - 1u);
    // nearest, ties to the earlier object (a mesh and a sphere are never the same object)
    if ((uint32_t)key != MISS && (tm < best.t || (tm == best.t && om < best.obj))) { best.obj = om; best.elem = em; best.t = tm; }
    if (best.obj != MISS) best.prim = sm[v.off_obj + best.obj * OBJ_STRIDE + OBJ_FIRST_PRIM] + best.elem;
    wave_sync();                                                      // rays/keys are rewritten by the next query
    return best;
}

template <int kAccel, class PS, class PB, bool kCount> struct Caster {
    static MCRT_DEV Hit run(PS sm, PB big, const SceneView& v, V3 o, V3 d, Counters& cn, bool live)
    {
        if (!live) { Hit h; h.prim = MISS; h.obj = MISS; h.elem = 0; h.t = kFltMax; return h; }
        if (kAccel == MCRT_ACCEL_BRUTE) return cast_brute<PS, PB, kCount>(sm, big, v, o, d, cn);
        return cast_octree<PS, PB, kCount>(sm, big, v, o, d, cn);
    }
};
// LDS words of a walk over an HBM-resident image, after the SMALL image: the stack columns, then the deal's per-thread hit keys (u64)
// and owner markers
constexpr uint32_t kWalkStackWords = LDS_STACK_ENTRIES * 2 * LDS_STACK_BLOCK;
constexpr uint32_t kWalkLdsWords = kWalkStackWords + 3 * LDS_STACK_BLOCK;
// Nearest hit of a ray batch (mcrt_intersect, mcrt_render_unlit) through an octree walked from HBM: the same dealt walk the render kernel
// runs, to the end of every ray.  All 64 lanes of the wavefront are here (the loops of those kernels are wave-uniform; a lane past the
// end of the batch has live == false and helps with the dealt records).
template <bool kCount> struct Caster<MCRT_ACCEL_OCTREE, LdsPtr, GlbPtr, kCount> {
    static MCRT_DEV Hit run(LdsPtr sm, GlbPtr big, const SceneView& v, V3 o, V3 d, Counters& cn, bool live)
    {
        typedef OctreeWalk<LdsPtr, GlbPtr, kCount> Walk;
        LdsRw base = (LdsRw)sm + ((v.small_words + 3u) & ~3u);
        LdsStack st;
        st.col = base + threadIdx.x;
        typename Walk::Deal L;
        L.key = (__attribute__((address_space(3))) unsigned long long*)(base + kWalkStackWords);
        L.slot = base + kWalkStackWords + 2 * LDS_STACK_BLOCK;
        L.slot[threadIdx.x] = 0u;          // the walk below starts its marker tags from scratch: nothing of an earlier batch may look like one
        wave_sync();
        Walk w;
        w.idle();
        if (live) w.begin_dealt(sm, big, v, st, L, o, d, cn);
        w.run_dealt(sm, big, v, st, L, o, d, cn, 0u);
        Hit h; h.prim = MISS; h.obj = MISS; h.elem = 0; h.t = kFltMax;
        if (live) { w.collect_dealt(sm, v, L); h = w.best; }
        wave_sync();                       // the keys are rewritten by the next batch
        return h;
    }
};

template <bool kCount> struct Caster<MCRT_ACCEL_FLAT, CstPtr, LdsPtr, kCount> {
    static MCRT_DEV Hit run(CstPtr sm, LdsPtr big, const SceneView& v, V3 o, V3 d, Counters& cn, bool live)
    {
        // the per-wavefront work area follows the staged triangle records in LDS
        LdsRw area = (LdsRw)(big + ((v.ntri * TRI_STRIDE + 3u) & ~3u)) + (threadIdx.x >> 6) * FLAT_AREA_WORDS;
        return cast_flat<kCount>(sm, big, area, v, o, d, cn, live);
    }
};
template <bool kCount> struct Caster<MCRT_ACCEL_SCAN, CstPtr, CstPtr, kCount> {
    static MCRT_DEV Hit run(CstPtr sm, CstPtr big, const SceneView& v, V3 o, V3 d, Counters& cn, bool live) { return cast_scan<kCount>(sm, big, v, o, d, cn, live); }
};

template <int kAccel, class PS, class PB, bool kCount>
MCRT_DEV Hit cast(PS sm, PB big, const SceneView& v, V3 o, V3 d, Counters& cn, bool live = true)
{
    // live == false: the lane takes part in the wavefront's cooperative work but has no ray of its own (answer: a miss)
    return Caster<kAccel, PS, PB, kCount>::run(sm, big, v, o, d, cn, live);
}
// --------------------------------------------------------------- materials
// Material::BRDF, source/core/material.h:24-53, for the (up to) two outgoing directions one diffuse hit needs - r0 towards the
// light sample (scene.cpp:137), r1 the hemisphere bounce (scene.cpp:164) - in one place.  The Oren-Nayar branch is by far the
// most expensive piece of shading (three acosf, sinf, tanf per direction pair, all through double-precision kernels), only a
// few lanes of a wavefront need it at a time, and a wavefront executes every copy of it that ANY lane needs: so it exists once.
// The three acosf run through one copy of the kernel (trip 0: theta_in, shared by both directions), the two directions through
// one copy of the sin/tan part.  A and B depend on the roughness alone (material.h:29-33) and come precomputed with the object
// record (same float operations, done once at upload).  Values are those of the reference's expression, operation by operation.
template <class PS>
MCRT_DEV void brdf2(PS sm, uint32_t orec, V3 incident, V3 normal, bool want0, V3 r0, bool want1, V3 r1, double& f0, double& f1)
{
    const float albedo = ldf(sm, orec + OBJ_ALBEDO);
    f0 = f1 = (double)albedo;
    if (sm[orec + OBJ_DIFFUSE] == MCRT_DIFFUSE_OREN_NAYAR) {
        const float A = ldf(sm, orec + OBJ_ON_A), B = ldf(sm, orec + OBJ_ON_B);
        const float cos_in = dot(incident, normal);
        const float cos_out0 = dot(r0, normal), cos_out1 = dot(r1, normal);
        const float cio0 = dot(incident, r0), cio1 = dot(incident, r1);
        float theta_in = 0.0f, theta_out0 = 0.0f, theta_out1 = 0.0f;
#pragma clang loop unroll(disable)
        for (int j = 0; j < 3; ++j) {
            if (j == 0 || (j == 1 && want0) || (j == 2 && want1)) {
                const float t = pm_acosf(j == 0 ? cos_in : (j == 1 ? cos_out0 : cos_out1));
                if (j == 0) theta_in = t; else if (j == 1) theta_out0 = t; else theta_out1 = t;
            }
        }
#pragma clang loop unroll(disable)
        for (int j = 0; j < 2; ++j) {
            if (j == 0 ? want0 : want1) {
                const float theta_out = j == 0 ? theta_out0 : theta_out1;
                const float cos_in_out = j == 0 ? cio0 : cio1;
                const float alpha = (theta_out < theta_in) ? theta_in : theta_out;       // std::max(theta_out, theta_in)
                const float beta = (theta_in < theta_out) ? theta_in : theta_out;        // std::min(theta_out, theta_in)
                const float mx = (0.0f < cos_in_out) ? cos_in_out : 0.0f;                 // std::max(0.0f, cos_in_out)
                float sin_alpha, tan_beta;
                pm_sin_and_tan(alpha, beta, sin_alpha, tan_beta);
                const double ON = (double)(A + (B * mx) * sin_alpha * tan_beta);
                const double f = (double)albedo * ON;
                if (j == 0) f0 = f; else f1 = f;
            }
        }
    }
}

// Scene::RandomHemisphereRay, source/scene.cpp:11-30
MCRT_DEV V3 hemisphere(V3 Ny, Rng& g, float& cosTheta)
{
    V3 Nx, Nz;
    if (fabsf(Ny.x) > fabsf(Ny.y)) Nx = mk(Ny.z, 0.0f, -Ny.x);
    else Nx = mk(0.0f, -Ny.z, Ny.y);
    Nx = normalize(Nx);
    Nz = normalize(cross(Ny, Nx));
    cosTheta = rng_float01(g);
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float phi = kTwoPiF * rng_float01(g);
    float sp, cp;
    pm_sincosf(phi, sp, cp);
    V3 s = mk(sinTheta * cp, cosTheta, sinTheta * sp);
    return mk(s.x * Nx.x + s.y * Ny.x + s.z * Nz.x, s.x * Nx.y + s.y * Ny.y + s.z * Nz.y, s.x * Nx.z + s.y * Ny.z + s.z * Nz.z);
}

// Object::GetRandomPointOnSurface: Light (source/objects/light.h:47-53), SphereObject
// (source/objects/sphere.h:59-69), base Object (source/objects/object.h:26-29).
template <class PS> MCRT_DEV V3 light_point(PS sm, uint32_t orec, Rng& g)
{
    V3 pos = ld3(sm, orec + OBJ_POS);
    uint32_t kind = sm[orec + OBJ_KIND];
    if (kind == MCRT_OBJECT_LIGHT) {
        float u = rng_float(g);
        float w = rng_float(g);
        V3 xv = ld3(sm, orec + OBJ_LIGHT_X), yv = ld3(sm, orec + OBJ_LIGHT_Y);
        V3 corner = (pos - vdivs(xv, 2.0f)) - vdivs(yv, 2.0f);
        return (corner + xv * u) + yv * w;
    }
    if (kind == MCRT_OBJECT_SPHERE) {
        float u = rng_float(g);
        float w = rng_float(g);
        float radius = ldf(sm, orec + OBJ_RADIUS);
        float theta = kTwoPiF * u;
        float phi = pm_acosf(2.0f * w - 1.0f);
        float sphi, cphi, sth, cth;
        pm_sincosf(phi, sphi, cphi);
        pm_sincosf(theta, sth, cth);
        return mk(pos.x + (radius * sphi * cth), pos.y + (radius * sphi * sth), pos.z + (radius * cphi));
    }
    return pos;
}

// Advance the generator past the draws light_point makes for this light, without sampling it.
template <class PS> MCRT_DEV void light_skip(PS sm, uint32_t orec, Rng& g)
{
    const uint32_t kind = sm[orec + OBJ_KIND];
    if (kind == MCRT_OBJECT_LIGHT || kind == MCRT_OBJECT_SPHERE) { (void)rng_next(g); (void)rng_next(g); }
}

// Object::GetSurfaceNormal: sphere.h:54-57 / mesh.cpp:43-46
template <class PS, class PB> MCRT_DEV V3 surface_normal(PS sm, PB big, const SceneView& v, const Hit& h, V3 p)
{
    uint32_t orec = v.off_obj + h.obj * OBJ_STRIDE;
    if (sm[orec + OBJ_KIND] == MCRT_OBJECT_SPHERE) return normalize(p - ld3(sm, orec + OBJ_POS));
    return ld3(big, v.off_tri + (sm[orec + OBJ_FIRST_TRI] + h.elem) * TRI_STRIDE + TRI_N);
}

// Scene::MaxImportance, source/scene.h:45-48 (std::max(a,b) = a<b ? b : a)
MCRT_DEV double max_importance(D3 i)
{
    double yz = (i.y < i.z) ? i.z : i.y;
    return (i.x < yz) ? yz : i.x;
}

// Camera::GetPixelRay, source/core/camera.h:36-57
struct CameraDev {
    float pos[3];
    float fov_scale;
    float view[16];
    uint32_t width, height;
    double dx, dy;       // PixelBuffer::deltaX/deltaY, source/core/pixelbuffer.cpp:20-21
    float aspect_f;      // float(pixels.aspectRatio())
};

MCRT_DEV void camera_ray(const CameraDev& c, float x, float y, V3& o, V3& d)
{
    V3 dir = mk((float)(-1.0f + x * c.dx), (float)(1.0f - y * c.dy), -1.0f);
    dir.x *= c.fov_scale * c.aspect_f;
    dir.y *= c.fov_scale;
    const float* m = c.view;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float add0 = m[0 + k] * dir.x + m[4 + k] * dir.y;
        float add1 = m[8 + k] * dir.z + m[12 + k] * 0.0f;
        r[k] = add0 + add1;
    }
    o = mk(c.pos[0], c.pos[1], c.pos[2]);
    d = normalize(mk(r[0], r[1], r[2]));
}

} // namespace mcrt
