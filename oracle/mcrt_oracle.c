/*
 * mcrt_oracle.c - TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C restatement of the reference renderer's CPU hot path
 * (DennyLindberg/Monte-Carlo-Ray-Tracer), function by function, with every
 * floating-point operation in the order the reference (and glm 0.9.9.2)
 * performs it.  Each function cites the reference lines it follows
 * (`source/...` is relative to the reference checkout).
 *
 * Pinning: tests/test_oracle_golden.py checks this file bit-for-bit against
 * fixtures under tests/golden/ that oracle/gen_golden.py produced by running
 * the reference's own translation units (oracle/_ref, built by
 * oracle/Makefile from /root/reference).  The reference ships no tests or
 * golden vectors of its own.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libmcrt_hip.so) never does.
 *
 * Build: gcc -std=gnu11 -O2 -ffp-contract=off (x86-64 baseline: SSE2 float
 * arithmetic, no FMA contraction, no excess precision).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mcrt.h"

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ options */
typedef struct orc_options {
    int32_t portable_math;  /* 0: libm sinf/cosf/tanf/acosf like the reference build;
                               1: the fixed-operation-order routines below, which the
                                  HIP kernels restate op for op (bit-exact GPU parity) */
    int32_t forward_sum;    /* 0: radiance summed in the reference's recursion order;
                               1: leaf/direct terms summed in depth-first visiting
                                  order, as the iterative GPU kernel does           */
} orc_options;

typedef struct orc_stats {
    uint64_t rays;           /* Scene::IntersectRay calls                        */
    uint64_t gates;          /* mesh AABB slab tests                             */
    uint64_t tri_tests;      /* Triangle::Intersects calls                       */
    uint64_t sphere_tests;   /* SphereObject::Intersects calls                   */
    uint64_t samples;        /* camera samples                                   */
    uint64_t trace_calls;    /* Scene::TraceRay calls                            */
    uint64_t splits;         /* refractive two-way splits                        */
    uint64_t max_pending;    /* most reflect branches waiting at once            */
    uint64_t max_depth_used; /* deepest recursion level reached                  */
    uint64_t nonfinite;      /* samples dropped for a non-finite component       */
} orc_stats;

/* ------------------------------------------------------------- vector math
 * glm semantics: dot = (x*x' + y*y') + z*z'
 * (include/glm/detail/func_geometric.inl:48-55), cross (:68-79),
 * normalize = v * (1/sqrt(dot)) (:82-90, func_exponential.inl:44-50),
 * reflect (:104-110). */
typedef struct { float x, y, z; } v3;
typedef struct { double x, y, z; } d3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdiv(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float vdot(v3 a, v3 b)
{
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
static inline v3 vcross(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 vnormalize(v3 v)
{
    float inv = 1.0f / sqrtf(vdot(v, v));
    return vmul(v, inv);
}
static inline v3 vreflect(v3 I, v3 N) { return vsub(I, vmul(vmul(N, vdot(N, I)), 2.0f)); }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

static inline d3 D(double x, double y, double z) { d3 r = { x, y, z }; return r; }
static inline d3 dadd(d3 a, d3 b) { return D(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline d3 dmul(d3 a, d3 b) { return D(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline d3 dscale(d3 a, double s) { return D(a.x * s, a.y * s, a.z * s); }
static inline d3 ddivs(d3 a, double s) { return D(a.x / s, a.y / s, a.z / s); }
static inline d3 ldd3(const double* p) { return D(p[0], p[1], p[2]); }

/* ------------------------------------------------------ portable float math
 * Used only when orc_options.portable_math != 0.  float functions evaluated
 * through double polynomials in a fixed operation order; the kernels in
 * csrc/ carry the same sequences, so device and host agree bit for bit.
 * Accuracy (checked in tests/test_host_and_abi.py::test_portable_trig_within_one_ulp_of_libm): within 1 ulp of libm,
 * equal to it on > 99.9 % of inputs.  Kernel coefficients are the classic
 * fdlibm minimax sets for sin/cos on [-pi/4, pi/4] and asin on [0, 0.5]. */
static const double PM_PIO2_HI = 1.57079632679489655800e+00;
static const double PM_PIO2_LO = 6.12323399573676603587e-17;
static const double PM_TWO_OVER_PI = 6.36619772367581382433e-01;
static const double PM_PI = 3.14159265358979311600e+00;

static inline double pm_ksin(double x)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double p = S5 + z * S6;
    p = S4 + z * p;
    p = S3 + z * p;
    p = S2 + z * p;
    p = S1 + z * p;
    return x + (x * z) * p;
}
static inline double pm_kcos(double x)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double p = C5 + z * C6;
    p = C4 + z * p;
    p = C3 + z * p;
    p = C2 + z * p;
    p = C1 + z * p;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
/* sin and cos of a float argument, as doubles (|x| up to a few thousand). */
static inline void pm_sincos_d(float xf, double* s, double* c)
{
    double x = (double)xf;
    double kd = floor(x * PM_TWO_OVER_PI + 0.5);
    double r = (x - kd * PM_PIO2_HI) - kd * PM_PIO2_LO;
    int q = (int)((long long)kd & 3);
    double sr = pm_ksin(r), cr = pm_kcos(r);
    double ss, cc;
    if (q == 0) { ss = sr; cc = cr; }
    else if (q == 1) { ss = cr; cc = -sr; }
    else if (q == 2) { ss = -sr; cc = -cr; }
    else { ss = -cr; cc = sr; }
    *s = ss; *c = cc;
}
static inline float pm_sinf(float x)
{
    if (!(fabsf(x) <= 8192.0f)) return (x - x) / (x - x); /* nan: inf/nan input, or outside the supported range */
    double s, c; pm_sincos_d(x, &s, &c); return (float)s;
}
static inline float pm_cosf(float x)
{
    if (!(fabsf(x) <= 8192.0f)) return (x - x) / (x - x);
    double s, c; pm_sincos_d(x, &s, &c); return (float)c;
}
static inline float pm_tanf(float x)
{
    if (!(fabsf(x) <= 8192.0f)) return (x - x) / (x - x);
    double s, c; pm_sincos_d(x, &s, &c); return (float)(s / c);
}
static inline double pm_asin_r(double z)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
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
static inline float pm_acosf(float xf)
{
    double x = (double)xf;
    if (!(fabs(x) <= 1.0)) return (float)((x - x) / (x - x)); /* nan, as acosf outside [-1,1] */
    if (fabs(x) < 0.5) {
        double z = x * x;
        double a = x + x * pm_asin_r(z);                 /* asin(x) */
        return (float)((PM_PIO2_HI - a) + PM_PIO2_LO);
    }
    if (x < 0.0) {
        double z = (1.0 + x) * 0.5;
        double s = sqrt(z);
        double a = s + s * pm_asin_r(z);                 /* asin(sqrt(z)) */
        return (float)(PM_PI - 2.0 * a);
    } else {
        double z = (1.0 - x) * 0.5;
        double s = sqrt(z);
        double a = s + s * pm_asin_r(z);
        return (float)(2.0 * a);
    }
}

/* exp / log / pow in double for the resolve step (source/main.cpp:150-161), fixed operation order (the classic
 * fdlibm argument reductions and minimax sets for log on [sqrt(2)/2, sqrt(2)) and exp on [-ln2/2, ln2/2]).  The
 * results are within 1 ulp (exp, log) and within ~1e-13 relative (pow = exp(y * log x)) of libm's: far more than the
 * 8-bit store needs - a byte differs from libm's only where 255 * pow lies within 1e-13 of an integer. */
static inline uint64_t pm_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double pm_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline double pm_log(double x)      /* finite x > 0 */
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10,
                 LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01, LG3 = 2.857142874366239149e-01,
                 LG4 = 2.222219843214978396e-01, LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
                 LG7 = 1.479819860511658591e-01;
    int64_t k = 0;
    uint64_t u = pm_bits(x);
    if (u < 0x0010000000000000ull) { x = x * 18014398509481984.0; u = pm_bits(x); k = -54; }     /* subnormal: scale by 2^54 */
    k += (int64_t)(u >> 52) - 1023;
    uint64_t m = u & 0x000FFFFFFFFFFFFFull;
    if (m >= 0x0006A09E667F3BCDull) { k += 1; m |= 0x3FE0000000000000ull; }                      /* mantissa in [sqrt(2)/2, 1) */
    else m |= 0x3FF0000000000000ull;                                                              /* mantissa in [1, sqrt(2)) */
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
static inline double pm_exp(double x)      /* any x */
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return pm_from_bits(0x7FF0000000000000ull);
    if (x < -7.45133219101941108420e+02) return 0.0;
    const int64_t k = (int64_t)(INV_LN2 * x + (x < 0.0 ? -0.5 : 0.5));
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
    /* y * 2^k: y is in [0.5, 2) */
    if (k >= -1021 && k <= 1022) return pm_from_bits(pm_bits(y) + ((uint64_t)k << 52));
    if (k > 1022) return pm_from_bits(pm_bits(y) + ((uint64_t)(k - 2) << 52)) * 4.0;
    return pm_from_bits(pm_bits(y) + ((uint64_t)(k + 1000) << 52)) * 9.33263618503218878990e-302;   /* 2^-1000 */
}
/* pow(x, y) for the tonemap: y = 1 / gamma.  Special cases as libm's pow: x^1 = x exactly, NaN in -> NaN, a negative base
 * with a non-integer exponent -> NaN (which std::min(1.0, c) then turns into 1.0, source/opengl/data.cpp:46). */
static inline double pm_pow(double x, double y)
{
    if (y == 1.0) return x;
    if (y == 0.0) return 1.0;
    if (x != x || y != y) return x + y;
    double sign = 1.0;
    if (x < 0.0) {
        if (y != floor(y)) return (x - x) / (x - x);
        if (fabs(y) < 9007199254740992.0 && floor(y * 0.5) * 2.0 != y) sign = -1.0;
        x = -x;
    }
    if (x == 0.0) return y > 0.0 ? 0.0 * sign : pm_from_bits(0x7FF0000000000000ull) * sign;
    if (x > 1.79769313486231570815e+308) return y > 0.0 ? x * sign : 0.0 * sign;
    return sign * pm_exp(y * pm_log(x));
}

ORC_EXPORT void orc_portable_exp_log_pow(int which, uint64_t n, const double* a, const double* b, double* out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = which == 0 ? pm_exp(a[i]) : which == 1 ? pm_log(a[i]) : pm_pow(a[i], b[i]);
}

typedef struct { float (*sinf_)(float); float (*cosf_)(float); float (*tanf_)(float); float (*acosf_)(float); } mathfn;
static const mathfn MATH_LIBM = { sinf, cosf, tanf, acosf };
static const mathfn MATH_PORTABLE = { pm_sinf, pm_cosf, pm_tanf, pm_acosf };

ORC_EXPORT void orc_portable_math(int which, uint64_t n, const float* in, float* out)
{
    for (uint64_t i = 0; i < n; ++i)
        out[i] = which == 0 ? pm_sinf(in[i]) : which == 1 ? pm_cosf(in[i]) : which == 2 ? pm_tanf(in[i]) : pm_acosf(in[i]);
}

/* ---------------------------------------------------------------------- rng
 * xorshift128+ and the mantissa trick, source/core/randomization.h:23-43;
 * wrappers source/core/randomization.cpp:15-33. */
typedef struct { uint64_t s[2]; } rng_t;

static inline uint64_t rng_next(rng_t* g)
{
    uint64_t x = g->s[0];
    const uint64_t y = g->s[1];
    g->s[0] = y;
    x ^= x << 23;
    g->s[1] = x ^ y ^ (x >> 17) ^ (y >> 26);
    return g->s[1] + y;
}
static inline double rng_double(rng_t* g)
{
    union { uint64_t i; double d; } u;
    u.i = (0x3FFull << 52) | (rng_next(g) >> 12);
    return u.d - 1.0;
}
static inline double rng_double_range(rng_t* g, double lo, double hi) { return lo + (hi - lo) * rng_double(g); }
static inline float rng_float(rng_t* g) { return (float)rng_double(g); }
static inline float rng_float_range(rng_t* g, float lo, float hi) { return lo + (hi - lo) * (float)rng_double(g); }

/* Counter -> initial state; the build's substitute for the reference's
 * std::random_device seeding (source/core/randomization.cpp:8-13).  Same
 * mapping as mcrt_rng_seed in the product and SampleSeed in ref_harness.cpp. */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline void sample_seed(uint64_t seed, uint32_t pixel, uint32_t sample, rng_t* g)
{
    uint64_t z = seed + (uint64_t)pixel * 0x9E3779B97F4A7C15ull + (uint64_t)sample * 0xD1B54A32D192ED03ull;
    g->s[0] = mix64(z + 0x9E3779B97F4A7C15ull);
    g->s[1] = mix64(z + 0x3C6EF372FE94F82Aull);
    if ((g->s[0] | g->s[1]) == 0) g->s[1] = 1;
}

ORC_EXPORT void orc_sample_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t out[2])
{
    rng_t g; sample_seed(seed, pixel, sample, &g); out[0] = g.s[0]; out[1] = g.s[1];
}
ORC_EXPORT void orc_rng_doubles(uint64_t s0, uint64_t s1, uint64_t n, double* out)
{
    rng_t g = { { s0, s1 } };
    for (uint64_t i = 0; i < n; ++i) out[i] = rng_double(&g);
}
ORC_EXPORT void orc_rng_floats(uint64_t s0, uint64_t s1, uint64_t n, float* out)
{
    rng_t g = { { s0, s1 } };
    for (uint64_t i = 0; i < n; ++i) out[i] = rng_float(&g);
}
ORC_EXPORT void orc_rng_floats_range(uint64_t s0, uint64_t s1, float lo, float hi, uint64_t n, float* out)
{
    rng_t g = { { s0, s1 } };
    for (uint64_t i = 0; i < n; ++i) out[i] = rng_float_range(&g, lo, hi);
}

/* -------------------------------------------------------------------- scene */
typedef struct orc_scene {
    uint32_t nobj, ntri, nlights;
    mcrt_object* obj;
    mcrt_triangle* tri;
    uint32_t* lights;        /* Scene::lights: indices of emissive objects, source/scene.cpp:41-49 */
    d3 background;
    struct orc_tree* tree;   /* optional: the product's flat octree, attached for the config-4 CPU baseline (see the replay section) */
} orc_scene;
struct orc_tree;
static void orc_tree_free(struct orc_tree* t);
ORC_EXPORT void orc_scene_detach_octree(orc_scene* s) { if (s->tree) { orc_tree_free(s->tree); s->tree = NULL; } }

ORC_EXPORT orc_scene* orc_scene_create(const mcrt_scene_desc* d)
{
    orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
    s->nobj = d->object_count; s->ntri = d->triangle_count;
    s->obj = (mcrt_object*)malloc(sizeof(mcrt_object) * (s->nobj ? s->nobj : 1));
    s->tri = (mcrt_triangle*)malloc(sizeof(mcrt_triangle) * (s->ntri ? s->ntri : 1));
    memcpy(s->obj, d->objects, sizeof(mcrt_object) * s->nobj);
    memcpy(s->tri, d->triangles, sizeof(mcrt_triangle) * s->ntri);
    s->lights = (uint32_t*)malloc(sizeof(uint32_t) * (s->nobj ? s->nobj : 1));
    for (uint32_t i = 0; i < s->nobj; ++i) {
        const mcrt_material* m = &s->obj[i].material;
        if (m->emission[0] > 0.0 || m->emission[1] > 0.0 || m->emission[2] > 0.0) s->lights[s->nlights++] = i;
    }
    s->background = ldd3(d->background);
    return s;
}
ORC_EXPORT void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    orc_scene_detach_octree(s);
    free(s->obj); free(s->tri); free(s->lights); free(s);
}
ORC_EXPORT uint32_t orc_scene_lights(const orc_scene* s, int32_t* out)
{
    for (uint32_t i = 0; i < s->nlights; ++i) out[i] = (int32_t)s->lights[i];
    return s->nlights;
}

/* Triangle ctor, source/core/triangle.h:17-23 */
ORC_EXPORT void orc_triangle_normal(const float v0[3], const float v1[3], const float v2[3], float out[3])
{
    v3 u = vsub(ld3(v1), ld3(v0));
    v3 v = vsub(ld3(v2), ld3(v0));
    st3(out, vnormalize(vcross(u, v)));
}

/* TriangleMesh::UpdateAABB, source/objects/mesh.cpp:55-68, with the AABB ctor
 * (source/core/aabb.h:17-21) and Encapsulate (:98-107). */
ORC_EXPORT void orc_mesh_aabb(const float position[3], const mcrt_triangle* tris, uint32_t n, float mn[3], float mx[3])
{
    v3 zero = V(0.0f, 0.0f, 0.0f);
    v3 half = vdiv(zero, 2.0f);
    v3 lo = vsub(ld3(position), half), hi = vadd(ld3(position), half);
    for (uint32_t i = 0; i < n; ++i) {
        const float* vs[3] = { tris[i].v0, tris[i].v1, tris[i].v2 };
        for (int k = 0; k < 3; ++k) {
            v3 p = ld3(vs[k]);
            lo.x = (p.x < lo.x) ? p.x : lo.x; lo.y = (p.y < lo.y) ? p.y : lo.y; lo.z = (p.z < lo.z) ? p.z : lo.z;
            hi.x = (hi.x < p.x) ? p.x : hi.x; hi.y = (hi.y < p.y) ? p.y : hi.y; hi.z = (hi.z < p.z) ? p.z : hi.z;
        }
    }
    st3(mn, lo); st3(mx, hi);
}

/* ------------------------------------------------------------------- camera
 * Camera ctor + SetView, source/core/camera.h:19-34;
 * glm::lookAtRH include/glm/ext/matrix_transform.inl:99-119;
 * glm::inverse(mat4) include/glm/detail/func_matrix.inl:294-352. */
static void mat4_inverse(const float m[4][4], float out[4][4])
{
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    float Vec0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] };
    float Vec1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    float Vec2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] };
    float Vec3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    static const float SignA[4] = { +1, -1, +1, -1 };
    static const float SignB[4] = { -1, +1, -1, +1 };
    float Inv[4][4];
    for (int i = 0; i < 4; ++i) {
        Inv[0][i] = ((Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i]) * SignA[i];
        Inv[1][i] = ((Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i]) * SignB[i];
        Inv[2][i] = ((Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i]) * SignA[i];
        Inv[3][i] = ((Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i]) * SignB[i];
    }
    float d0 = m[0][0] * Inv[0][0], d1 = m[0][1] * Inv[1][0], d2 = m[0][2] * Inv[2][0], d3_ = m[0][3] * Inv[3][0];
    float Dot1 = (d0 + d1) + (d2 + d3_);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) out[c][r] = Inv[c][r] * OneOverDeterminant;
}

ORC_EXPORT void orc_camera_make(uint32_t w, uint32_t h, float fov_y, const float pos[3], const float target[3], const float up[3], mcrt_camera* out)
{
    float halfAngle = (fov_y * 0.5f);
    float radians = halfAngle / 180.0f * (float)3.14159265358979323846264338327950288;
    out->fov_scale = tanf(radians);
    v3 eye = ld3(pos), center = ld3(target), upv = ld3(up);
    v3 f = vnormalize(vsub(center, eye));
    v3 s = vnormalize(vcross(f, upv));
    v3 u = vcross(s, f);
    float m[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    m[0][0] = s.x; m[1][0] = s.y; m[2][0] = s.z;
    m[0][1] = u.x; m[1][1] = u.y; m[2][1] = u.z;
    m[0][2] = -f.x; m[1][2] = -f.y; m[2][2] = -f.z;
    m[3][0] = -vdot(s, eye); m[3][1] = -vdot(u, eye); m[3][2] = vdot(f, eye);
    float inv[4][4];
    mat4_inverse(m, inv);
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out->view[c * 4 + r] = inv[c][r];
    st3(out->position, eye);
    out->width = w; out->height = h;
}

/* Camera::GetPixelRay, source/core/camera.h:36-57 (deltaX/deltaY/aspect are
 * doubles, source/core/pixelbuffer.cpp:20-22; mat4*vec4 ordering
 * include/glm/detail/type_mat4x4.inl:561-572). */
static inline void camera_ray(const mcrt_camera* c, float x, float y, v3* o, v3* d)
{
    double dx = 2.0 / (double)c->width;
    double dy = 2.0 / (double)c->height;
    double aspect = c->width / (double)c->height;
    const float xOrigin = -1.0f, yOrigin = 1.0f;
    v3 dir = V((float)(xOrigin + x * dx), (float)(yOrigin - y * dy), -1.0f);
    dir.x *= c->fov_scale * (float)aspect;
    dir.y *= c->fov_scale;
    const float* m = c->view;
    float v[4] = { dir.x, dir.y, dir.z, 0.0f };
    float r[3];
    for (int k = 0; k < 3; ++k) {
        float add0 = m[0 * 4 + k] * v[0] + m[1 * 4 + k] * v[1];
        float add1 = m[2 * 4 + k] * v[2] + m[3 * 4 + k] * v[3];
        r[k] = add0 + add1;
    }
    *o = ld3(c->position);
    *d = vnormalize(V(r[0], r[1], r[2]));
}
ORC_EXPORT void orc_camera_rays(const mcrt_camera* c, uint64_t n, const float* xs, const float* ys, float* out_o, float* out_d)
{
    for (uint64_t i = 0; i < n; ++i) {
        v3 o, d; camera_ray(c, xs[i], ys[i], &o, &d);
        st3(out_o + 3 * i, o); st3(out_d + 3 * i, d);
    }
}

/* ------------------------------------------------------------- intersection */
typedef struct { int32_t obj; uint32_t elem; float t; } hit_t;

/* AABB::IntersectsRay, source/core/aabb.h:49-96.  t1/t2 are float quotients
 * widened to double; min/max are std::min/std::max on doubles. */
static inline int aabb_hit(const float mn[3], const float mx[3], v3 o, v3 d)
{
    double tmin = -(double)FLT_MAX, tmax = (double)FLT_MAX, t1, t2, lo, hi;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0) {
            t1 = (mn[a] - oo[a]) / dd[a];
            t2 = (mx[a] - oo[a]) / dd[a];
            lo = (t2 < t1) ? t2 : t1;                 /* std::min(t1,t2) */
            hi = (t1 < t2) ? t2 : t1;                 /* std::max(t1,t2) */
            tmin = (tmin < lo) ? lo : tmin;           /* std::max(tmin, lo) */
            tmax = (hi < tmax) ? hi : tmax;           /* std::min(tmax, hi) */
        }
    }
    return tmax >= tmin;
}

/* Triangle::Intersects, source/core/triangle.h:25-62 */
static inline int tri_hit(const mcrt_triangle* T, v3 o, v3 d, float* t)
{
    v3 v0 = ld3(T->v0);
    v3 edge1 = vsub(ld3(T->v1), v0);
    v3 edge2 = vsub(ld3(T->v2), v0);
    v3 h = vcross(d, edge2);
    float a = vdot(edge1, h);
    if (fabsf(a) < FLT_EPSILON) return 0;
    float f = 1.0f / a;
    v3 s = vsub(o, v0);
    float u = f * vdot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = vcross(s, edge1);
    float v = f * vdot(d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    *t = f * vdot(edge2, q);
    return *t > FLT_EPSILON;
}

/* SphereObject::Intersects, source/objects/sphere.h:16-52 */
static inline int sphere_hit(const mcrt_object* S, v3 o, v3 d, float* t)
{
    float radius = S->radius;
    if (radius < FLT_EPSILON) return 0;
    float radiusSq = radius * radius;
    float t0, t1;
    v3 L = vsub(ld3(S->position), o);
    float tca = vdot(L, d);
    if (tca < 0) return 0;
    float distanceSq = vdot(L, L) - tca * tca;
    if (distanceSq > radiusSq) return 0;
    float thc = sqrtf(radiusSq - distanceSq);
    t0 = tca - thc;
    t1 = tca + thc;
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0) {
        t0 = t1;
        if (t0 < 0) return 0;
    }
    *t = t0;
    return 1;
}

/* Scene::IntersectRay, source/scene.cpp:62-80, with TriangleMesh::Intersects,
 * source/objects/mesh.cpp:9-41. */
static hit_t scene_hit_octree(const orc_scene* S, v3 o, v3 d, orc_stats* st);
static inline hit_t scene_hit_brute(const orc_scene* S, v3 o, v3 d, orc_stats* st)
{
    hit_t best = { -1, 0, FLT_MAX };
    if (st) st->rays++;
    for (uint32_t i = 0; i < S->nobj; ++i) {
        const mcrt_object* ob = &S->obj[i];
        if (ob->kind == MCRT_OBJECT_SPHERE) {
            float t;
            if (st) st->sphere_tests++;
            if (sphere_hit(ob, o, d, &t) && t < best.t) { best.obj = (int32_t)i; best.elem = 0; best.t = t; }
        } else {
            if (st) st->gates++;
            if (!aabb_hit(ob->aabb_min, ob->aabb_max, o, d)) continue;
            uint32_t elementIndex = 0;
            float nearest = FLT_MAX, hitDistance = FLT_MAX;
            for (uint32_t k = 0; k < ob->triangle_count; ++k) {
                if (st) st->tri_tests++;
                if (tri_hit(&S->tri[ob->first_triangle + k], o, d, &hitDistance) && hitDistance < nearest) {
                    nearest = hitDistance; elementIndex = k;
                }
            }
            if (nearest < FLT_MAX && nearest < best.t) { best.obj = (int32_t)i; best.elem = elementIndex; best.t = nearest; }
        }
    }
    return best;
}
static inline hit_t scene_hit(const orc_scene* S, v3 o, v3 d, orc_stats* st)
{
    if (S->tree) return scene_hit_octree(S, o, d, st);
    return scene_hit_brute(S, o, d, st);
}

ORC_EXPORT void orc_intersect(const orc_scene* S, uint64_t n, const float* o, const float* d, int32_t* out_obj, uint32_t* out_elem,
                              float* out_t, orc_stats* st)
{
    for (uint64_t i = 0; i < n; ++i) {
        hit_t h = scene_hit(S, ld3(o + 3 * i), ld3(d + 3 * i), st);
        out_obj[i] = h.obj; out_elem[i] = h.elem; out_t[i] = h.t;
    }
}

/* ------------------------------------------------------- flat-octree replay
 * CPU replay of the product's flat octree traversal (cast_octree in
 * monte-carlo-ray-tracer_amd/csrc/mcrt_device.h; layout in csrc/mcrt_layout.h)
 * with visit counters.  The structure itself is product data (downloaded with
 * mcrt_scene_image_octree / mcrt_octree_download); the leaf tests are this
 * file's restatements, so `replay == orc_intersect` proves the structure can
 * never hide the nearest hit, and the counters define the algorithmic bytes of
 * SURVEY.md 8(d).  orc_scene_attach_octree makes scene_hit - and with it
 * TraceRay and the render loop - query through the replay: the "restatement's
 * octree path" that is the CPU baseline of BASELINE config 4, where the
 * reference's brute force (10^6 triangle tests per ray) is not feasible. */
typedef struct orc_octree {
    uint32_t node_count;
    const float* box;        /* node_count * 6: min xyz, max xyz (tight, padded)     */
    const uint32_t* words;   /* node_count * 2: child word (mask<<24 | first child), first prim ref */
    const uint32_t* refs;    /* primitive ids, bit 31 = last of the node's list      */
    float root_cell[6];      /* mcrt_octree_info: root_cell_min, root_cell_max       */
    float pad;               /* mcrt_octree_info: pad                                */
    const float* cell;       /* node_count * 4: cell min xyz, 2^-depth (orc_octree_cells) */
} orc_octree;

typedef struct orc_octree_stats { uint64_t rays, nodes, tris, spheres, gates; } orc_octree_stats;
#ifdef ORC_WALK_DEBUG
/* experiment builds only (tools/exp_walk_stats.py): where the steps of a walk go */
ORC_EXPORT uint64_t orc_walk_dbg[32];
#define WALK_DBG(i, n) (orc_walk_dbg[i] += (uint64_t)(n))
#else
#define WALK_DBG(i, n) ((void)0)
#endif

struct orc_tree {
    orc_octree view;
    float* box; uint32_t* words; uint32_t* refs; float* cell;   /* owned copies (attach) */
    uint32_t* first_prim; uint32_t* prim_obj;         /* primitive ids are assigned in (object, element) order */
};

/* Every node's cell from the root cell, the way the builder halves cells (csrc/mcrt_flatten.cpp: centre = min + 0.5f * (max - min),
 * octant bit a set = upper half on axis a): the values the node records carry (NODE_CELL, NODE_SCALE). */
ORC_EXPORT void orc_octree_cells(const orc_octree* T, float* out)
{
    float* mx = (float*)malloc(sizeof(float) * 3 * (size_t)T->node_count);
    for (int a = 0; a < 3; ++a) { out[a] = T->root_cell[a]; mx[a] = T->root_cell[3 + a]; }
    out[3] = 1.0f;
    for (uint32_t n = 0; n < T->node_count; ++n) {      /* breadth-first order: a parent comes before its children */
        const uint32_t cw = T->words[2 * (size_t)n], base = cw & 0x00FFFFFFu, mask = cw >> 24;
        uint32_t rank = 0;
        for (uint32_t c = 0; c < 8; ++c) {
            if (!(mask & (1u << c))) continue;
            const uint32_t ch = base + rank++;
            for (int a = 0; a < 3; ++a) {
                const float lo = out[4 * (size_t)n + a], hi = mx[3 * (size_t)n + a];
                const float half = 0.5f * (hi - lo), centre = lo + half;
                out[4 * (size_t)ch + a] = ((c >> a) & 1u) ? centre : lo;
                mx[3 * (size_t)ch + a] = ((c >> a) & 1u) ? hi : centre;
            }
            out[4 * (size_t)ch + 3] = out[4 * (size_t)n + 3] * 0.5f;
        }
    }
    free(mx);
}

static void prim_tables(const orc_scene* S, uint32_t** first_prim_out, uint32_t** prim_obj_out)
{
    uint32_t nprim = 0;
    uint32_t* first_prim = (uint32_t*)malloc(sizeof(uint32_t) * (S->nobj + 1));
    for (uint32_t i = 0; i < S->nobj; ++i) { first_prim[i] = nprim; nprim += (S->obj[i].kind == MCRT_OBJECT_SPHERE) ? 1u : S->obj[i].triangle_count; }
    first_prim[S->nobj] = nprim;
    uint32_t* prim_obj = (uint32_t*)malloc(sizeof(uint32_t) * (nprim ? nprim : 1));
    for (uint32_t i = 0; i < S->nobj; ++i) for (uint32_t p = first_prim[i]; p < first_prim[i + 1]; ++p) prim_obj[p] = i;
    *first_prim_out = first_prim; *prim_obj_out = prim_obj;
}

/* OctreeWalk::visit of csrc/mcrt_device.h, step by step: pop, the node's tight box, the octants the ray passes from the
 * cell's faces and mid planes (no child record is read), then the node's own list. */
static hit_t octree_hit(const orc_scene* S, const orc_octree* T, const uint32_t* first_prim, const uint32_t* prim_obj, v3 ro, v3 rd,
                        orc_octree_stats* st, int* overflow)
{
    enum { STACK = 128 };
    uint64_t gate_known = 0, gate_pass = 0;
    /* OctreeWalk::begin: an origin further out than far_limit = pad / (32 * FLT_EPSILON) (or not a number) is answered by the loop
       over all objects - out there an ulp of the origin's coordinate is no longer small against the padding the walk decides with */
    {
        const float far_limit = T->pad * 262144.0f;
        if (!(fabsf(ro.x) <= far_limit && fabsf(ro.y) <= far_limit && fabsf(ro.z) <= far_limit)) return scene_hit_brute(S, ro, rd, NULL);
    }
    /* slab_rcp of csrc/mcrt_device.h: the reciprocal clamped to +-2^100 so that every product below stays finite */
    const float RCP_MAX = 1.2676506002282294e30f;
    float inv[3] = { 1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z }, padt[3];
    const float dd[3] = { rd.x, rd.y, rd.z };
    uint32_t flip = 0;
    for (int a = 0; a < 3; ++a) {
        if (!(fabsf(inv[a]) <= RCP_MAX)) inv[a] = copysignf(RCP_MAX, dd[a]);
        padt[a] = (0.5f * T->pad) * fabsf(inv[a]);      /* SceneView::walk_pad: HALF the builder's padding, see OctreeWalk::visit */
        if (inv[a] < 0.0f) flip |= 1u << a;
    }
    const float oo[3] = { ro.x, ro.y, ro.z };
    const float root_size[3] = { T->root_cell[3] - T->root_cell[0], T->root_cell[4] - T->root_cell[1], T->root_cell[5] - T->root_cell[2] };
    hit_t best = { -1, 0, FLT_MAX };
    uint32_t bestPrim = 0xFFFFFFFFu;
    uint32_t stack_node[STACK]; float stack_t[STACK];
    int sp = 0;
    stack_node[sp] = 0; stack_t[sp] = 0.0f; ++sp;
    if (st) st->rays++;
    while (sp > 0) {
        --sp;
        uint32_t node = stack_node[sp];
        if (stack_t[sp] > best.t) { WALK_DBG(0, 1); continue; }      /* popped behind the best hit */
        if (st) st->nodes++;
        float tb0 = 0.0f, tb1 = best.t;         /* the stretch of the ray inside the node's tight box */
        {
            const float* b = T->box + 6 * (size_t)node;
            for (int a = 0; a < 3; ++a) {
                float x = (b[a] - oo[a]) * inv[a], y = (b[3 + a] - oo[a]) * inv[a];
                tb0 = fmaxf(tb0, fminf(x, y)); tb1 = fminf(tb1, fmaxf(x, y));
            }
            if (!(tb0 <= tb1)) { WALK_DBG(1, 1); continue; }         /* fetched, tight box missed */
        }
        uint32_t cw = T->words[2 * (size_t)node];
        WALK_DBG(2, 1);
        uint32_t base = cw & 0x00FFFFFFu, mask = cw >> 24;
        if (mask) {
            /* OctreeWalk::visit, csrc/mcrt_device.h: the octant the ray is in at tb0 plus one per mid plane crossed before tb1, in ray
               order (bit a of r set = beyond the mid plane of axis a), every decision padded on the generous side */
            const float* cell = T->cell + 4 * (size_t)node;
            uint32_t r0 = 0;
            float k[3]; uint32_t bit[3] = { 1u, 2u, 4u };
            for (int a = 0; a < 3; ++a) {
                const float tm = fmaf(root_size[a] * inv[a] * cell[3], 0.5f, (cell[a] - oo[a]) * inv[a]);
                const float enter = tm - padt[a];
                const int behind = enter < tb0 - 2.0f * padt[a];
                if (behind) r0 |= 1u << a;
                k[a] = (!behind && enter <= tb1) ? enter : INFINITY;
            }
#define ORC_CSWAP(i, j) if (k[j] < k[i]) { float t_ = k[i]; k[i] = k[j]; k[j] = t_; uint32_t u_ = bit[i]; bit[i] = bit[j]; bit[j] = u_; }
            ORC_CSWAP(0, 1) ORC_CSWAP(1, 2) ORC_CSWAP(0, 1)
#undef ORC_CSWAP
            const uint32_t r[4] = { r0, r0 ^ bit[0], r0 ^ bit[0] ^ bit[1], r0 ^ bit[0] ^ bit[1] ^ bit[2] };
            for (int j = 3; j >= 0; --j) {
                const float t_in = j ? fmaxf(tb0, k[j - 1]) : tb0;
                if (j && !(k[j - 1] <= best.t)) continue;
                const uint32_t oct = r[j] ^ flip;
                if (!(mask & (1u << oct))) continue;
                const uint32_t child = base + (uint32_t)__builtin_popcount(mask & ((1u << oct) - 1u));
                if (sp < STACK) { stack_node[sp] = child; stack_t[sp] = t_in; ++sp; } else *overflow = 1;
                WALK_DBG(4, 1);
            }
            WALK_DBG(3, 1);                                          /* interior visits */
        }
        uint32_t ref = T->words[2 * (size_t)node + 1];
        if (ref != 0xFFFFFFFFu) {
            uint32_t e;
            WALK_DBG(5, 1);                                          /* visits with a list */
            do {
                WALK_DBG(6, 1);
                e = T->refs[ref++];
                uint32_t prim = e & 0x7FFFFFFFu;
                uint32_t ob = prim_obj[prim], el = prim - first_prim[ob];
                const mcrt_object* O = &S->obj[ob];
                float t = 0.0f; int h;
                int better;
                if (O->kind == MCRT_OBJECT_SPHERE) { if (st) st->spheres++; h = sphere_hit(O, ro, rd, &t); }
                else {
                    int pass;
                    if (ob >= 64u) {          /* beyond the 64-entry gate cache: evaluated on every contact */
                        if (st) st->gates++;
                        pass = aabb_hit(O->aabb_min, O->aabb_max, ro, rd);
                    } else {
                        uint64_t bit = 1ull << ob;
                        if (!(gate_known & bit)) {
                            if (st) st->gates++;
                            gate_known |= bit;
                            if (aabb_hit(O->aabb_min, O->aabb_max, ro, rd)) gate_pass |= bit;
                        }
                        pass = (gate_pass & bit) != 0;
                    }
                    h = 0;
                    if (pass) { if (st) st->tris++; h = tri_hit(&S->tri[O->first_triangle + el], ro, rd, &t); }
                }
                better = h && (t < best.t || (t == best.t && bestPrim != 0xFFFFFFFFu && prim < bestPrim));
                if (better) { best.obj = (int32_t)ob; best.elem = el; best.t = t; bestPrim = prim; }
            } while (!(e & 0x80000000u));
        }
    }
    return best;
}

ORC_EXPORT int orc_intersect_octree(const orc_scene* S, const orc_octree* T, uint64_t n, const float* o, const float* d,
                                    int32_t* out_obj, uint32_t* out_elem, float* out_t, orc_octree_stats* st)
{
    uint32_t *first_prim, *prim_obj;
    prim_tables(S, &first_prim, &prim_obj);
    int overflow = 0;
    for (uint64_t i = 0; i < n; ++i) {
        hit_t best = octree_hit(S, T, first_prim, prim_obj, ld3(o + 3 * i), ld3(d + 3 * i), st, &overflow);
        out_obj[i] = best.obj; out_elem[i] = best.elem; out_t[i] = best.t;
    }
    free(first_prim); free(prim_obj);
    return overflow;
}

/* From here on every nearest-hit query of this scene (orc_intersect, TraceRay, render) walks the given flat octree
 * instead of the object list.  Same hits by construction (tests); different cost.  The arrays are copied. */
ORC_EXPORT void orc_scene_attach_octree(orc_scene* S, const orc_octree* T, uint32_t ref_count)
{
    struct orc_tree* tr = (struct orc_tree*)calloc(1, sizeof *tr);
    size_t nb = sizeof(float) * 6 * (size_t)T->node_count, nw = sizeof(uint32_t) * 2 * (size_t)T->node_count, nr = sizeof(uint32_t) * (size_t)(ref_count ? ref_count : 1);
    tr->box = (float*)malloc(nb); tr->words = (uint32_t*)malloc(nw); tr->refs = (uint32_t*)malloc(nr);
    memcpy(tr->box, T->box, nb); memcpy(tr->words, T->words, nw); memcpy(tr->refs, T->refs, sizeof(uint32_t) * (size_t)ref_count);
    tr->cell = (float*)malloc(sizeof(float) * 4 * (size_t)T->node_count);
    tr->view = *T;
    tr->view.box = tr->box; tr->view.words = tr->words; tr->view.refs = tr->refs; tr->view.cell = tr->cell;
    if (T->cell) memcpy(tr->cell, T->cell, sizeof(float) * 4 * (size_t)T->node_count);
    else orc_octree_cells(&tr->view, tr->cell);
    prim_tables(S, &tr->first_prim, &tr->prim_obj);
    orc_scene_detach_octree(S);
    S->tree = tr;
}

static hit_t scene_hit_octree(const orc_scene* S, v3 o, v3 d, orc_stats* st)
{
    int overflow = 0;
    orc_octree_stats os = { 0, 0, 0, 0, 0 };
    hit_t h = octree_hit(S, &S->tree->view, S->tree->first_prim, S->tree->prim_obj, o, d, st ? &os : NULL, &overflow);
    if (st) { st->rays += os.rays; st->gates += os.gates; st->tri_tests += os.tris; st->sphere_tests += os.spheres; }
    return h;
}
static void orc_tree_free(struct orc_tree* t)
{
    free(t->box); free(t->words); free(t->refs); free(t->cell); free(t->first_prim); free(t->prim_obj); free(t);
}

/* ---------------------------------------------------------------- materials */
/* Material::BRDF, source/core/material.h:24-53 */
static inline double brdf(const mcrt_material* m, v3 incident, v3 reflection, v3 normal, const mathfn* mf)
{
    if (m->diffuse == MCRT_DIFFUSE_OREN_NAYAR) {
        float sigma2 = m->roughness * m->roughness;
        float A = 1.0f - 0.5f * sigma2 / (sigma2 + 0.57f);
        float B = 0.45f * sigma2 / (sigma2 + 0.09f);
        float cos_in = vdot(incident, normal);
        float cos_out = vdot(reflection, normal);
        float cos_in_out = vdot(incident, reflection);
        float theta_in = mf->acosf_(cos_in);
        float theta_out = mf->acosf_(cos_out);
        float alpha = (theta_out < theta_in) ? theta_in : theta_out;   /* glm::max(theta_out, theta_in) */
        float beta = (theta_in < theta_out) ? theta_in : theta_out;    /* glm::min(theta_out, theta_in) */
        float mx = (0.0f < cos_in_out) ? cos_in_out : 0.0f;            /* glm::max(0.0f, cos_in_out)    */
        double ON = (A + (B * mx) * mf->sinf_(alpha) * mf->tanf_(beta));
        return m->albedo * ON;
    }
    return m->albedo;
}
ORC_EXPORT void orc_brdf(const mcrt_material* m, uint64_t n, const float* incident, const float* reflection, const float* normal,
                         int portable_math, double* out)
{
    const mathfn* mf = portable_math ? &MATH_PORTABLE : &MATH_LIBM;
    for (uint64_t i = 0; i < n; ++i) out[i] = brdf(m, ld3(incident + 3 * i), ld3(reflection + 3 * i), ld3(normal + 3 * i), mf);
}

/* Scene::RandomHemisphereRay, source/scene.cpp:11-30 */
static inline v3 hemisphere(v3 Ny, rng_t* g, float* cosTheta, const mathfn* mf)
{
    v3 Nx, Nz;
    if (fabsf(Ny.x) > fabsf(Ny.y)) Nx = V(Ny.z, 0, -Ny.x);
    else Nx = V(0, -Ny.z, Ny.y);
    Nx = vnormalize(Nx);
    Nz = vnormalize(vcross(Ny, Nx));
    *cosTheta = rng_float_range(g, 0.0f, 1.0f);
    float sinTheta = sqrtf(1 - *cosTheta * *cosTheta);
    float phi = (float)6.28318530717958647692528676655900576 * rng_float_range(g, 0.0f, 1.0f);
    v3 sample = V(sinTheta * mf->cosf_(phi), *cosTheta, sinTheta * mf->sinf_(phi));
    return V(sample.x * Nx.x + sample.y * Ny.x + sample.z * Nz.x,
             sample.x * Nx.y + sample.y * Ny.y + sample.z * Nz.y,
             sample.x * Nx.z + sample.y * Ny.z + sample.z * Nz.z);
}
ORC_EXPORT void orc_hemisphere(uint64_t n, const float* normals, const uint64_t* states, int portable_math, float* out_dir, float* out_cos)
{
    const mathfn* mf = portable_math ? &MATH_PORTABLE : &MATH_LIBM;
    for (uint64_t i = 0; i < n; ++i) {
        rng_t g = { { states[2 * i], states[2 * i + 1] } };
        float c;
        st3(out_dir + 3 * i, hemisphere(ld3(normals + 3 * i), &g, &c, mf));
        out_cos[i] = c;
    }
}

/* Object::GetSurfaceNormal: SphereObject source/objects/sphere.h:54-57,
 * TriangleMesh source/objects/mesh.cpp:43-46 */
static inline v3 surface_normal(const orc_scene* S, int32_t obj, uint32_t elem, v3 p)
{
    const mcrt_object* ob = &S->obj[obj];
    if (ob->kind == MCRT_OBJECT_SPHERE) return vnormalize(vsub(p, ld3(ob->position)));
    return ld3(S->tri[ob->first_triangle + elem].normal);
}
ORC_EXPORT void orc_surface_normal(const orc_scene* S, uint64_t n, const int32_t* obj, const uint32_t* elem, const float* pts, float* out)
{
    for (uint64_t i = 0; i < n; ++i) st3(out + 3 * i, surface_normal(S, obj[i], elem[i], ld3(pts + 3 * i)));
}

/* Object::GetRandomPointOnSurface: Light source/objects/light.h:47-53,
 * SphereObject source/objects/sphere.h:59-69, base Object
 * source/objects/object.h:26-29. */
static inline v3 light_point(const mcrt_object* L, rng_t* g, const mathfn* mf)
{
    v3 pos = ld3(L->position);
    if (L->kind == MCRT_OBJECT_LIGHT) {
        float u = rng_float(g);
        float v = rng_float(g);
        v3 xv = ld3(L->light_x), yv = ld3(L->light_y);
        v3 corner = vsub(vsub(pos, vdiv(xv, 2.0f)), vdiv(yv, 2.0f));
        return vadd(vadd(corner, vmul(xv, u)), vmul(yv, v));
    }
    if (L->kind == MCRT_OBJECT_SPHERE) {
        float u = rng_float(g);
        float v = rng_float(g);
        float theta = (float)6.28318530717958647692528676655900576 * u;
        float phi = mf->acosf_(2.0f * v - 1.0f);
        float x = pos.x + (L->radius * mf->sinf_(phi) * mf->cosf_(theta));
        float y = pos.y + (L->radius * mf->sinf_(phi) * mf->sinf_(theta));
        float z = pos.z + (L->radius * mf->cosf_(phi));
        return V(x, y, z);
    }
    return pos;
}

/* Object::GetRandomPointOnSurface for one object, one point per generator state (advanced state written back) */
ORC_EXPORT void orc_object_random_points(const orc_scene* S, uint32_t index, uint64_t n, uint64_t* states, int portable_math, float* out)
{
    const mathfn* mf = portable_math ? &MATH_PORTABLE : &MATH_LIBM;
    for (uint64_t i = 0; i < n; ++i) {
        rng_t g = { { states[2 * i], states[2 * i + 1] } };
        st3(out + 3 * i, light_point(&S->obj[index], &g, mf));
        states[2 * i] = g.s[0]; states[2 * i + 1] = g.s[1];
    }
}

/* Object::Intersects of ONE object (source/objects/object.h:23): SphereObject::Intersects, source/objects/sphere.h:16-52, or
 * TriangleMesh::Intersects, source/objects/mesh.cpp:9-41; hitInfo starts reset (source/core/ray.h:17-29) */
ORC_EXPORT void orc_object_intersects(const orc_scene* S, uint32_t index, uint64_t n, const float* o, const float* d, uint8_t* out_hit,
                                      uint32_t* out_elem, float* out_t)
{
    const mcrt_object* ob = &S->obj[index];
    for (uint64_t i = 0; i < n; ++i) {
        v3 ro = ld3(o + 3 * i), rd = ld3(d + 3 * i);
        int hit = 0; uint32_t elem = 0; float t = FLT_MAX;
        if (ob->kind == MCRT_OBJECT_SPHERE) { float ts; if (sphere_hit(ob, ro, rd, &ts)) { hit = 1; t = ts; } }
        else if (aabb_hit(ob->aabb_min, ob->aabb_max, ro, rd)) {
            float nearest = FLT_MAX, hitDistance = FLT_MAX;
            for (uint32_t k = 0; k < ob->triangle_count; ++k)
                if (tri_hit(&S->tri[ob->first_triangle + k], ro, rd, &hitDistance) && hitDistance < nearest) { nearest = hitDistance; elem = k; }
            if (nearest < FLT_MAX) { hit = 1; t = nearest; } else elem = 0;
        }
        out_hit[i] = (uint8_t)hit; out_elem[i] = elem; out_t[i] = t;
    }
}

/* ------------------------------------------------------------------ TraceRay
 * Scene::TraceRay, source/scene.cpp:94-234.  `pending` counts reflect
 * branches that are waiting while a refract branch runs (the refract operand
 * of the `+` at :212-213 is evaluated first by g++; fixture F5 pins that). */
typedef struct {
    const orc_scene* S;
    const mathfn* mf;
    orc_stats* st;
    int forward;
    d3 fwd;                  /* forward_sum accumulator */
    uint64_t pending;
    uint32_t depth0;
} trace_ctx;

static inline double max_importance(d3 i)
{
    double yz = (i.y < i.z) ? i.z : i.y;      /* std::max(y, z)        */
    return (i.x < yz) ? yz : i.x;             /* std::max(x, max(y,z)) */
}
static inline void fwd_add(trace_ctx* C, d3 v) { if (C->forward) C->fwd = dadd(C->fwd, v); }

static d3 trace_ray(trace_ctx* C, v3 ro, v3 rd, rng_t* g, unsigned traceDepth, d3 importance)
{
    const orc_scene* S = C->S;
    const float MARGIN = FLT_EPSILON * 20.0f;
    if (C->st) {
        C->st->trace_calls++;
        uint64_t lvl = (uint64_t)(C->depth0 - traceDepth);
        if (lvl > C->st->max_depth_used) C->st->max_depth_used = lvl;
    }
    hit_t hit = scene_hit(S, ro, rd, C->st);
    if (hit.obj < 0) { d3 r = dmul(importance, S->background); fwd_add(C, r); return r; }

    const mcrt_object* object = &S->obj[hit.obj];
    if (traceDepth == 0 || object->kind == MCRT_OBJECT_LIGHT) {
        d3 r = dmul(importance, ldd3(object->material.emission)); fwd_add(C, r); return r;
    }

    const mcrt_material* surface = &object->material;
    v3 intersectionPoint = vadd(ro, vmul(rd, hit.t));
    v3 normal = surface_normal(S, hit.obj, hit.elem, intersectionPoint);
    if (surface->surface == MCRT_SURFACE_DIFFUSE) {
        intersectionPoint = vadd(intersectionPoint, vmul(normal, MARGIN));

        d3 directLight = D(0.0, 0.0, 0.0);
        float surfaceDot = 0.0f, lightDot = 0.0f;
        double BRDF = 0.0;
        for (uint32_t li = 0; li < S->nlights; ++li) {
            const mcrt_object* lightSource = &S->obj[S->lights[li]];
            v3 lightDirection = vsub(light_point(lightSource, g, C->mf), intersectionPoint);
            lightDirection = vnormalize(lightDirection);
            hit_t sh = scene_hit(S, intersectionPoint, lightDirection, C->st);
            if (sh.obj < 0 || (uint32_t)sh.obj == S->lights[li]) {
                surfaceDot = vdot(normal, lightDirection);
                lightDot = vdot(V(0.0f, -1.0f, 0.0f), vmul(lightDirection, -1.0f));
                BRDF = brdf(surface, rd, lightDirection, normal, C->mf);
                directLight = dadd(directLight, dscale(ldd3(lightSource->material.emission), (double)(BRDF * surfaceDot * lightDot)));
            }
        }

        double p = max_importance(importance);
        importance = dmul(importance, ldd3(surface->color));
        directLight = dmul(directLight, importance);
        fwd_add(C, directLight);
        if (rng_double_range(g, 0.0, 1.0) > p) return directLight;
        importance = ddivs(importance, p);

        v3 bounce = hemisphere(normal, g, &surfaceDot, C->mf);
        importance = dscale(importance, 2.0 * (double)surfaceDot * brdf(surface, rd, bounce, normal, C->mf));
        d3 indirectLight = trace_ray(C, intersectionPoint, bounce, g, traceDepth - 1, importance);
        return dadd(directLight, indirectLight);
    } else if (surface->surface == MCRT_SURFACE_SPECULAR) {
        intersectionPoint = vadd(intersectionPoint, vmul(normal, MARGIN));
        v3 newDirection = vreflect(rd, normal);
        return trace_ray(C, intersectionPoint, newDirection, g, traceDepth - 1, importance);
    } else if (surface->surface == MCRT_SURFACE_REFRACTIVE) {
        v3 I = rd;
        float n1 = 1.0f;
        float n2 = surface->refractive_index;
        if (vdot(normal, I) >= 0) {
            normal = vmul(normal, -1.0f);
            float tmp = n1; n1 = n2; n2 = tmp;
        }
        v3 errorMargin = vmul(normal, MARGIN);
        float n = n1 / n2;
        float cosI = vdot(I, normal);
        float cos2t = 1.0f - n * n * (1.0f - cosI * cosI);
        if (cos2t < 0.0f)
            return trace_ray(C, vadd(intersectionPoint, errorMargin), vreflect(I, normal), g, traceDepth - 1, importance);

        v3 tdir = vsub(vmul(I, n), vmul(normal, (cosI * n + sqrtf(cos2t))));
        float R0 = (n2 - n1) / (n2 + n1);
        R0 *= R0;
        float c = 1.0f - (-cosI);
        float R = R0 + (1.0f - R0) * c * c * c * c * c;

        if (rng_double(g) < max_importance(importance)) {
            if (C->st) {
                C->st->splits++;
                if (C->pending + 1 > C->st->max_pending) C->st->max_pending = C->pending + 1;
            }
            C->pending++;
            d3 refr = trace_ray(C, vsub(intersectionPoint, errorMargin), tdir, g, traceDepth - 1, dscale(importance, (double)(1.0 - R)));
            C->pending--;
            d3 refl = trace_ray(C, vadd(intersectionPoint, errorMargin), vreflect(I, normal), g, traceDepth - 1, dscale(importance, (double)R));
            return dadd(refl, refr);
        } else {
            double P = .25 + .5 * R;
            if (rng_double(g) < P) {
                importance = dscale(importance, R / P);
                return trace_ray(C, vadd(intersectionPoint, errorMargin), vreflect(I, normal), g, traceDepth - 1, importance);
            } else {
                importance = dscale(importance, (1.0 - R) / (1.0 - P));
                return trace_ray(C, vsub(intersectionPoint, errorMargin), tdir, g, traceDepth - 1, importance);
            }
        }
    }
    return D(0.0, 0.0, 0.0);
}

static inline d3 trace_top(const orc_scene* S, v3 o, v3 d, rng_t* g, uint32_t depth, const orc_options* op, orc_stats* st)
{
    trace_ctx C;
    C.S = S; C.mf = (op && op->portable_math) ? &MATH_PORTABLE : &MATH_LIBM; C.st = st;
    C.forward = op ? op->forward_sum : 0; C.fwd = D(0.0, 0.0, 0.0); C.pending = 0; C.depth0 = depth;
    d3 r = trace_ray(&C, o, d, g, depth, D(1.0, 1.0, 1.0));
    return C.forward ? C.fwd : r;
}

ORC_EXPORT void orc_trace_rays(const orc_scene* S, uint64_t n, const float* o, const float* d, const uint64_t* states, uint32_t depth,
                               const orc_options* op, double* out, orc_stats* st)
{
    for (uint64_t i = 0; i < n; ++i) {
        rng_t g = { { states[2 * i], states[2 * i + 1] } };
        d3 c = trace_top(S, ld3(o + 3 * i), ld3(d + 3 * i), &g, depth, op, st);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

/* One iteration of the spp loop, source/main.cpp:122-139 */
static inline d3 one_sample(const orc_scene* S, const mcrt_camera* cam, uint32_t pixel, uint32_t sample, uint64_t seed, uint32_t depth,
                            const orc_options* op, orc_stats* st)
{
    rng_t g; sample_seed(seed, pixel, sample, &g);
    uint32_t x = pixel % cam->width, y = pixel / cam->width;
    float sx = rng_float(&g);
    float sy = rng_float(&g);
    v3 o, d; camera_ray(cam, (float)x + sx, (float)y + sy, &o, &d);
    if (st) st->samples++;
    return trace_top(S, o, d, &g, depth, op, st);
}

ORC_EXPORT void orc_trace_samples(const orc_scene* S, const mcrt_camera* cam, uint64_t n, const uint32_t* pixels, const uint32_t* samples,
                                  uint64_t seed, uint32_t depth, const orc_options* op, double* out, orc_stats* st)
{
    for (uint64_t i = 0; i < n; ++i) {
        d3 c = one_sample(S, cam, pixels[i], samples[i], seed, depth, op, st);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

/* --------------------------------------------------------------- full render
 * Headless version of TracePixels (source/main.cpp:169-184) in sequential
 * mode (atomic next-pixel counter, :89-95), PixelBuffer::Accumulate
 * (source/core/pixelbuffer.cpp:39-46) with non-finite samples dropped. */
typedef struct {
    const orc_scene* S; const mcrt_camera* cam; uint32_t s0, sn, depth; uint64_t seed; const orc_options* op;
    double* sums; uint32_t npix; uint32_t* next; orc_stats st; int want_stats; uint32_t lanes;
} render_job;

static void* render_worker(void* arg)
{
    render_job* J = (render_job*)arg;
    for (;;) {
        uint32_t p = __atomic_fetch_add(J->next, 1u, __ATOMIC_RELAXED);
        if (p >= J->npix) break;
        /* `lanes` sub-slices summed separately and then combined in order mirror
         * the GPU kernel's lanes-per-pixel split; lanes == 1 is plain sample order. */
        double tot[3] = { 0.0, 0.0, 0.0 };
        uint32_t per = J->sn / J->lanes;
        for (uint32_t l = 0; l < J->lanes; ++l) {
            double r = 0.0, g = 0.0, b = 0.0;
            uint32_t b0 = l * per, b1 = (l + 1 == J->lanes) ? J->sn : b0 + per;
            for (uint32_t k = b0; k < b1; ++k) {
                d3 v = one_sample(J->S, J->cam, p, J->s0 + k, J->seed, J->depth, J->op, J->want_stats ? &J->st : NULL);
                if (isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) { r += v.x; g += v.y; b += v.z; }
                else J->st.nonfinite++;
            }
            if (l == 0) { tot[0] = r; tot[1] = g; tot[2] = b; } else { tot[0] += r; tot[1] += g; tot[2] += b; }
        }
        J->sums[3 * (size_t)p] += tot[0]; J->sums[3 * (size_t)p + 1] += tot[1]; J->sums[3 * (size_t)p + 2] += tot[2];
    }
    return NULL;
}

ORC_EXPORT uint64_t orc_render(const orc_scene* S, const mcrt_camera* cam, uint32_t sample_begin, uint32_t sample_count, uint32_t depth,
                               uint64_t seed, uint32_t threads, uint32_t lanes_per_pixel, const orc_options* op, double* sums, orc_stats* st)
{
    if (threads == 0) threads = 1;
    if (threads > 256) threads = 256;
    if (lanes_per_pixel == 0 || lanes_per_pixel > sample_count) lanes_per_pixel = 1;
    uint32_t next = 0;
    render_job jobs[256];
    pthread_t tid[256];
    for (uint32_t t = 0; t < threads; ++t) {
        render_job* J = &jobs[t];
        memset(J, 0, sizeof *J);
        J->S = S; J->cam = cam; J->s0 = sample_begin; J->sn = sample_count; J->depth = depth; J->seed = seed; J->op = op;
        J->sums = sums; J->npix = cam->width * cam->height; J->next = &next; J->want_stats = st != NULL; J->lanes = lanes_per_pixel;
        if (t > 0) pthread_create(&tid[t], NULL, render_worker, J);
    }
    render_worker(&jobs[0]);
    uint64_t dropped = 0;
    for (uint32_t t = 0; t < threads; ++t) {
        if (t > 0) pthread_join(tid[t], NULL);
        dropped += jobs[t].st.nonfinite;
        if (st) {
            const orc_stats* a = &jobs[t].st;
            st->rays += a->rays; st->gates += a->gates; st->tri_tests += a->tri_tests; st->sphere_tests += a->sphere_tests;
            st->samples += a->samples; st->trace_calls += a->trace_calls; st->splits += a->splits; st->nonfinite += a->nonfinite;
            if (a->max_pending > st->max_pending) st->max_pending = a->max_pending;
            if (a->max_depth_used > st->max_depth_used) st->max_depth_used = a->max_depth_used;
        }
    }
    return dropped;
}

/* Scene::TraceUnlit, source/scene.cpp:82-92, via pixel centres, source/main.cpp:126 */
ORC_EXPORT void orc_render_unlit(const orc_scene* S, const mcrt_camera* cam, double* out)
{
    for (uint32_t y = 0; y < cam->height; ++y)
        for (uint32_t x = 0; x < cam->width; ++x) {
            v3 o, d; camera_ray(cam, (float)x + 0.5f, (float)y + 0.5f, &o, &d);
            hit_t h = scene_hit(S, o, d, NULL);
            size_t i = 3 * ((size_t)y * cam->width + x);
            if (h.obj >= 0) {
                const double* c = S->obj[h.obj].material.color;
                out[i] = c[0]; out[i + 1] = c[1]; out[i + 2] = c[2];
            } else { out[i] = 0.0f; out[i + 1] = 0.0f; out[i + 2] = 0.0f; }
        }
}

/* Mean + tonemap + 8-bit BGRA store: source/main.cpp:142-164,
 * GLImageBuffer::SetPixel source/opengl/data.cpp:28-50. */
/* portable_math != 0: pow / exp through the fixed-order pm_pow / pm_exp (what k_resolve runs), else libm like the reference. */
ORC_EXPORT void orc_resolve_ex(uint64_t npix, const double* sums, uint64_t spp, int tonemap, double gamma, double exposure, int portable_math,
                               uint8_t* out_bgra, double* out_mean)
{
    for (uint64_t p = 0; p < npix; ++p) {
        double c[3];
        for (int k = 0; k < 3; ++k) {
            c[k] = sums[3 * p + k] / (double)spp;
            if (out_mean) out_mean[3 * p + k] = c[k];
            if (tonemap == MCRT_TONEMAP_REINHARD) {
                c[k] = c[k] / (c[k] + 1.0);
                c[k] = portable_math ? pm_pow(c[k], 1.0 / gamma) : pow(c[k], 1.0 / gamma);
            } else if (tonemap == MCRT_TONEMAP_EXPOSURE) {
                c[k] = 1.0 - (portable_math ? pm_exp(-c[k] * exposure) : exp(-c[k] * exposure));
                c[k] = portable_math ? pm_pow(c[k], 1.0 / gamma) : pow(c[k], 1.0 / gamma);
            }
            double t = (c[k] < 1.0) ? c[k] : 1.0;    /* std::min(1.0, c) */
            t = (t < 0.0) ? 0.0 : t;                 /* std::max(t, 0.0) */
            c[k] = t;
        }
        if (out_bgra) {
            out_bgra[4 * p] = (uint8_t)(c[2] * 255.0);
            out_bgra[4 * p + 1] = (uint8_t)(c[1] * 255.0);
            out_bgra[4 * p + 2] = (uint8_t)(c[0] * 255.0);
            out_bgra[4 * p + 3] = (uint8_t)(1.0 * 255.0);
        }
    }
}

ORC_EXPORT void orc_resolve(uint64_t npix, const double* sums, uint64_t spp, int tonemap, double gamma, double exposure, uint8_t* out_bgra,
                            double* out_mean)
{
    orc_resolve_ex(npix, sums, spp, tonemap, gamma, exposure, 0, out_bgra, out_mean);
}

ORC_EXPORT uint32_t orc_abi_sizes(uint32_t out[4])
{
    out[0] = (uint32_t)sizeof(mcrt_material); out[1] = (uint32_t)sizeof(mcrt_triangle);
    out[2] = (uint32_t)sizeof(mcrt_object); out[3] = (uint32_t)sizeof(mcrt_camera);
    return (uint32_t)sizeof(orc_stats);
}
