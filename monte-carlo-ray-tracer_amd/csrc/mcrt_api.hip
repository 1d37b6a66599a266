// mcrt_api.hip - kernels and the C ABI of libmcrt_hip.so (include/mcrt.h).
//
// Kernel family (all hand-written for gfx950, 64-wide wavefronts):
//   k_render        the hot loop of source/main.cpp:100-184 + Scene::TraceRay
//                   (source/scene.cpp:94-234): persistent workgroups pull work
//                   items (sample slice, pixel) from an atomic counter; a lane
//                   renders the samples of its item in order; every loop
//                   iteration is exactly one nearest-hit query (camera/bounce
//                   ray or shadow ray), so the wave stays converged on the
//                   expensive part while lanes sit in different path states.
//   k_combine       adds the slice sums of a pass to the accumulators in slice
//                   order (images too small to fill the GPU with one lane per pixel).
//   k_trace_list    the same per-sample code on an explicit (pixel, sample) list
//                   (parity entry point mcrt_trace_samples).
//   k_intersect     Scene::IntersectRay (source/scene.cpp:62-80) on a ray batch.
//   k_camera_rays   Camera::GetPixelRay (source/core/camera.h:36-57).
//   k_unlit         Scene::TraceUnlit (source/scene.cpp:82-92).
//   k_resolve       mean / tonemap / gamma / BGRA8 (source/main.cpp:142-164,
//                   source/opengl/data.cpp:28-55).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mcrt.h"
#include "mcrt_device.h"
#include "mcrt_flatten.h"
#include "mcrt_build.h"

using namespace mcrt;

namespace {

constexpr int kBlock = 256;                 // 4 wavefronts
constexpr uint32_t kPendFields = 16;        // words per parked reflect branch: o(3) d(3) depth(1) pad(1) importance(6) pad(2) = one 64-byte line
constexpr uint32_t kMaxDepthLimit = 256;    // bound on max_depth (size of the parked-branch stack)
constexpr uint32_t kLdsBudgetBytes = 96 * 1024;  // whole-scene LDS residency threshold

// ------------------------------------------------------------------ staging
// Copy the scene image(s) into LDS.  Returns the LDS view of SMALL; BIG follows it when resident.
template <bool kBigLds>
__device__ __forceinline__ void stage_scene(const SceneView& v, uint32_t* smem)
{
    for (uint32_t i = threadIdx.x; i < v.small_words; i += blockDim.x) smem[i] = v.small_img[i];
    if (kBigLds) {
        uint32_t base = (v.small_words + 3u) & ~3u;
        for (uint32_t i = threadIdx.x; i < v.big_words; i += blockDim.x) smem[base + i] = v.big_img[i];
    }
    __syncthreads();
}

template <bool kBigLds> struct BigPtr;
template <> struct BigPtr<true> { typedef LdsPtr type; };
template <> struct BigPtr<false> { typedef GlbPtr type; };

template <bool kBigLds>
__device__ __forceinline__ typename BigPtr<kBigLds>::type big_ptr(const SceneView& v, uint32_t* smem);
template <> __device__ __forceinline__ LdsPtr big_ptr<true>(const SceneView& v, uint32_t* smem)
{
    return (LdsPtr)(smem + ((v.small_words + 3u) & ~3u));
}
template <> __device__ __forceinline__ GlbPtr big_ptr<false>(const SceneView& v, uint32_t*) { return v.big_img; }

// Where a kernel reads the scene images from: LDS copies (octree / brute), or straight from
// global memory through the constant address space (scan: wave-uniform addresses -> scalar loads).
template <bool kBigLds, int kAccel> struct SceneMem {
    typedef LdsPtr PS;
    typedef typename BigPtr<kBigLds>::type PB;
    static __device__ __forceinline__ void open(const SceneView& v, uint32_t* smem, PS& sm, PB& big)
    {
        stage_scene<kBigLds>(v, smem);
        sm = (LdsPtr)smem;
        big = big_ptr<kBigLds>(v, smem);
    }
};
// flat scan: object records through scalar loads, triangle records from an LDS copy (the view these kernels get has
// off_tri == 0: flat_view() on the host), nothing else of the images is staged
template <bool kBigLds> struct SceneMem<kBigLds, MCRT_ACCEL_FLAT> {
    typedef CstPtr PS;
    typedef LdsPtr PB;
    static __device__ __forceinline__ void open(const SceneView& v, uint32_t* smem, PS& sm, PB& big)
    {
        const uint32_t words = v.ntri * TRI_STRIDE;
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) smem[i] = v.big_img[v.tri_src + i];
        __syncthreads();
        sm = (CstPtr)v.small_img;
        big = (LdsPtr)smem;
    }
};
template <bool kBigLds> struct SceneMem<kBigLds, MCRT_ACCEL_SCAN> {
    typedef CstPtr PS;
    typedef CstPtr PB;
    static __device__ __forceinline__ void open(const SceneView& v, uint32_t*, PS& sm, PB& big)
    {
        sm = (CstPtr)v.small_img;
        big = (CstPtr)v.big_img;
    }
};

// ------------------------------------------------------------ parked branches
// Reflect branches of refractive splits (source/scene.cpp:212-213) wait here while the refract
// branch runs.  One column per launched lane; fields interleaved so a wave touches consecutive words.
struct PendStack {
    uint32_t* base;
    uint32_t entries;  // capacity per lane (= max_depth)
    uint32_t lane;     // this lane's global index
    // one 64-byte line per (lane, entry): a push or pop touches exactly one line
    __device__ __forceinline__ U4* line(uint32_t entry) const
    {
        return reinterpret_cast<U4*>(base + ((size_t)lane * entries + entry) * kPendFields);
    }
    __device__ __forceinline__ void push(uint32_t entry, V3 o, V3 d, D3 imp, uint32_t depth) const
    {
        U4* p = line(entry);
        const uint64_t ix = (uint64_t)__double_as_longlong(imp.x), iy = (uint64_t)__double_as_longlong(imp.y), iz = (uint64_t)__double_as_longlong(imp.z);
        U4 a, b, c, e;
        a.x = __float_as_uint(o.x); a.y = __float_as_uint(o.y); a.z = __float_as_uint(o.z); a.w = __float_as_uint(d.x);
        b.x = __float_as_uint(d.y); b.y = __float_as_uint(d.z); b.z = depth; b.w = 0;
        c.x = (uint32_t)ix; c.y = (uint32_t)(ix >> 32); c.z = (uint32_t)iy; c.w = (uint32_t)(iy >> 32);
        e.x = (uint32_t)iz; e.y = (uint32_t)(iz >> 32); e.z = 0; e.w = 0;
        p[0] = a; p[1] = b; p[2] = c; p[3] = e;
    }
    __device__ __forceinline__ void pop(uint32_t entry, V3& o, V3& d, D3& imp, uint32_t& depth) const
    {
        const U4* p = line(entry);
        const U4 a = p[0], b = p[1], c = p[2], e = p[3];
        o = mk(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        d = mk(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
        depth = b.z;
        imp = dmk(__longlong_as_double((long long)((uint64_t)c.x | ((uint64_t)c.y << 32))), __longlong_as_double((long long)((uint64_t)c.z | ((uint64_t)c.w << 32))),
                  __longlong_as_double((long long)((uint64_t)e.x | ((uint64_t)e.y << 32))));
    }
};

__device__ __forceinline__ U4 u4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { U4 r; r.x = a; r.y = b; r.z = c; r.w = d; return r; }
__device__ __forceinline__ uint32_t dlo(double v) { return (uint32_t)(uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ uint32_t dhi(double v) { return (uint32_t)((uint64_t)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ double dmake(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32))); }
__device__ __forceinline__ U4 dd4(double a, double b) { return u4(dlo(a), dhi(a), dlo(b), dhi(b)); }

struct RenderCounters { uint32_t samples, trace_calls, splits; };

template <class PB> struct WalkStackFor {
    typedef ScratchStack type;
    static constexpr bool kDeal = false;
    static __device__ __forceinline__ void bind(type&, uint32_t*) {}
};
template <> struct WalkStackFor<GlbPtr> {
    typedef LdsStack type;
    static constexpr bool kDeal = true;       // leaf records dealt over the wavefront (OctreeWalk::run_dealt)
    static __device__ __forceinline__ void bind(type& st, uint32_t* lds) { st.col = (__attribute__((address_space(3))) uint32_t*)lds + threadIdx.x; }
};
// LDS words the kernels that trace paths need after the SMALL image (octree walk over an HBM-resident image): the stack columns, then the
// deal's per-thread hit keys (u64), hit objects and owner markers
template <class W> __device__ __forceinline__ typename W::Deal walk_deal_area(uint32_t* walk_lds)
{
    typename W::Deal L;
    L.key = (__attribute__((address_space(3))) unsigned long long*)(walk_lds + kWalkStackWords);
    L.slot = (LdsRw)(walk_lds + kWalkStackWords + 2 * LDS_STACK_BLOCK);
    return L;
}
static_assert(LDS_STACK_BLOCK == kBlock, "LdsStack columns are one per thread of the block");

// Scheduling knobs of trace_lane (wave-uniform runtime values, so that they can be swept without rebuilding):
//   min_trace   lanes with a camera / bounce ray are served when at least this many are waiting (0: always)
//   fetch_min   idle lanes fetch their next items together once this many are idle (1: at once)
//   refill      octree walk: hand back control once fewer than refill/64 of the lanes that entered are still walking (0: walk to the end)
struct LanePolicy { uint32_t min_trace, fetch_min, refill; };

enum : uint32_t { ST_NEW = 0, ST_TRACE = 1, ST_SHADOW = 2, ST_DONE = 3, ST_FETCH = 4 };

// Explicit start of a TraceRay call (mcrt_trace_rays): ray, generator state and importance come from
// the caller instead of the camera / counter seed; the advanced generator state is written back.
struct ExplicitStart {
    const float* o;
    const float* d;
    uint64_t* state;
    const double* importance;   // may be null: (1,1,1)
};

// One lane of a tracing kernel.  `work` hands out items - samples [s_begin, s_begin + s_count) of a
// pixel - one at a time (Work::next) and takes back the sum of the item's finite samples in sample
// order plus the number dropped (Work::done; keep_nonfinite: the raw value, used by the parity path
// with s_count == 1).  Lanes fetch on their own: a lane that finishes early does not wait for its
// wavefront.
//
// A wavefront executes every piece of code that ANY of its lanes needs in an iteration, and with 64 lanes in 64
// different path states that is nearly all of it, every iteration: the cost of the loop is its static size.  So the
// shading of a diffuse hit is laid out to exist once: everything that follows from the hit alone - the light sample,
// the roulette draw, the hemisphere bounce, the BRDF towards the light AND towards the bounce (brdf2: one copy of the
// Oren-Nayar trigonometry serves both) - is computed when the hit is shaded, in the reference's draw order
// (scene.cpp:127-164: light points, roulette, hemisphere; the shadow ray consumes no random numbers, so its outcome
// cannot move a draw).  What waits for the shadow query is one comparison and one multiply-add.
// kMulti: scenes with more than one light test the further lights one query at a time from a saved copy of the
// generator (the state that path needs is compiled out of the one-light kernels).
// The per-lane state of the estimator between two nearest-hit queries, and the two halves of one loop iteration around the
// query: pre() - take the next work item / start the next camera sample, so that the lane has a ray - and post() - shade the
// hit (source/scene.cpp:94-234).  The render kernels run pre / query / post in one loop per lane (trace_lane); the wavefront
// pipeline (k_wf_shade / k_wf_trace, big scenes) keeps the same state in HBM between kernels and runs the same two functions,
// so both produce the same numbers by construction.
template <bool kMulti> struct LaneState {
    D3 sum;
    uint32_t dropped;
    uint32_t pixel, s_begin, s_count;
    uint32_t state;
    uint32_t k;
    Rng g;
    V3 ro, rd;                  // ray of the current query
    D3 imp, L;
    uint32_t depth, npend;
    // diffuse-hit context kept across the shadow queries
    V3 hp, bd;                  // hit point (origin of the shadow rays and of the bounce), bounce direction
    D3 impc;                    // importance * surface colour, scene.cpp:148
    D3 direct;
    double p_survive, g_bounce; // roulette threshold; 2 * cosTheta * BRDF(bounce), scene.cpp:164
    double w_light;             // BRDF(light) * surfaceDot * lightDot of the light under test, scene.cpp:136-139
    bool survive;
    uint32_t light_i;
    // kMulti only: what sampling and shading a further light needs
    V3 hn, hd;
    uint32_t horec;
    Rng g_l;

    __device__ __forceinline__ void init()
    {
        sum = dmk(0.0, 0.0, 0.0); dropped = 0; pixel = s_begin = s_count = 0; state = ST_FETCH; k = 0;
        g.s0 = 0; g.s1 = 0; ro = mk(0, 0, 0); rd = mk(0, 0, 1); imp = dmk(1, 1, 1); L = dmk(0, 0, 0); depth = 0; npend = 0;
        hp = mk(0, 0, 0); bd = mk(0, 0, 0); impc = dmk(0, 0, 0); direct = dmk(0, 0, 0); p_survive = 1.0; g_bounce = 0.0; w_light = 0.0;
        survive = false; light_i = 0; hn = mk(0, 0, 0); hd = mk(0, 0, 0); horec = 0; g_l.s0 = 0; g_l.s1 = 0;
    }

    // Before the query: a lane without an item takes the next one (when `fetch_ok`), a lane between samples starts the next sample.
    // (no early return on the fetch path: in the render kernels every lane has to arrive at the cooperative query that follows)
    template <class Work>
    __device__ __forceinline__ void pre(bool fetch_ok, Work& work, const CameraDev& cam, uint64_t seed, uint32_t max_depth, ExplicitStart& exs,
                                        const ExplicitStart*& ex, RenderCounters& rc, bool count)
    {
        if (state == ST_FETCH && fetch_ok) {
            if (!work.next(pixel, s_begin, s_count)) state = ST_DONE;
            else {
                ex = work.explicit_start(exs) ? &exs : nullptr;
                k = 0; sum = dmk(0.0, 0.0, 0.0); dropped = 0;
                state = ST_NEW;
            }
        }
        if (state == ST_NEW) {
            // one iteration of the spp loop, source/main.cpp:122-139
            imp = dmk(1.0, 1.0, 1.0);
            if (ex) {
                g.s0 = ex->state[0]; g.s1 = ex->state[1];
                ro = mk(ex->o[0], ex->o[1], ex->o[2]);
                rd = mk(ex->d[0], ex->d[1], ex->d[2]);
                if (ex->importance) imp = dmk(ex->importance[0], ex->importance[1], ex->importance[2]);
            } else {
                uint64_t a, b;
                sample_seed(seed, pixel, s_begin + k, a, b);
                g.s0 = a; g.s1 = b;
                float sx = rng_float(g);
                float sy = rng_float(g);
                const uint32_t px = pixel % cam.width, py = pixel / cam.width;
                camera_ray(cam, (float)px + sx, (float)py + sy, ro, rd);
            }
            L = dmk(0.0, 0.0, 0.0);
            depth = max_depth;
            npend = 0;
            state = ST_TRACE;
            if (count) rc.samples++;
        }
    }

    // After the query.  A wavefront executes every piece of code that ANY of its lanes needs in an iteration, and with 64 lanes in 64
    // different path states that is nearly all of it, every iteration: the cost of the loop is its static size.  So the
    // shading of a diffuse hit is laid out to exist once: everything that follows from the hit alone - the light sample,
    // the roulette draw, the hemisphere bounce, the BRDF towards the light AND towards the bounce (brdf2: one copy of the
    // Oren-Nayar trigonometry serves both) - is computed when the hit is shaded, in the reference's draw order
    // (scene.cpp:127-164: light points, roulette, hemisphere; the shadow ray consumes no random numbers, so its outcome
    // cannot move a draw).  What waits for the shadow query is one comparison and one multiply-add.
    // kMulti: scenes with more than one light test the further lights one query at a time from a saved copy of the
    // generator (the state that path needs is compiled out of the one-light kernels).
    // `lazy.item(*this)` is called before the first access to the item's fields (sum, pixel, s_begin, s_count, k, dropped): a caller that
    // keeps the state parked in memory fetches them only for the lanes whose sample ends (trace_lane_parked); NoLazy for everybody else.
    struct NoLazy { __device__ __forceinline__ void item(LaneState&) const {} };
    template <class PS, class PB, class Work>
    __device__ __forceinline__ void post(const Hit& hit, PS sm, PB big, const SceneView& v, const D3& background, const PendStack& pend, bool keep_nonfinite,
                                         Work& work, const ExplicitStart* ex, RenderCounters& rc, bool count)
    {
        NoLazy lazy;
        post(hit, sm, big, v, background, pend, keep_nonfinite, work, ex, rc, count, lazy);
    }
    template <class PS, class PB, class Work, class Lazy>
    __device__ __forceinline__ void post(const Hit& hit, PS sm, PB big, const SceneView& v, const D3& background, const PendStack& pend, bool keep_nonfinite,
                                         Work& work, const ExplicitStart* ex, RenderCounters& rc, bool count, Lazy& lazy)
    {
        bool branch_done = false;      // this TraceRay chain returned; continue with a parked branch or finish the sample
        bool lights_done = false;      // every light of the current diffuse hit is accounted for
        bool eval_light = false, eval_bounce = false;   // BRDF values wanted this iteration: towards `ldir`, towards `bd`
        V3 ldir = mk(0, 0, 0), en = mk(0, 0, 0), ed = mk(0, 0, 0);   // light direction; normal and incident direction of the hit being shaded
        uint32_t eorec = 0;
        float cosTheta = 0.0f;
        if (state == ST_TRACE) {
            if (count) rc.trace_calls++;
            if (hit.obj == MISS) {
                L = L + imp * background;                                   // scene.cpp:97-100
                branch_done = true;
            } else {
                const uint32_t orec = v.off_obj + hit.obj * OBJ_STRIDE;
                const uint32_t kind = sm[orec + OBJ_KIND];
                if (depth == 0 || kind == MCRT_OBJECT_LIGHT) {
                    L = L + imp * ldd3(sm, orec + OBJ_EMISSION);            // scene.cpp:103-107
                    branch_done = true;
                } else {
                    V3 p = ro + rd * hit.t;                                 // scene.cpp:111
                    V3 n = surface_normal(sm, big, v, hit, p);              // scene.cpp:112
                    const uint32_t surface = sm[orec + OBJ_SURFACE];
                    if (surface == MCRT_SURFACE_DIFFUSE) {
                        p = p + n * kMargin;                                // scene.cpp:115
                        hp = p; en = n; ed = rd; eorec = orec;
                        direct = dmk(0.0, 0.0, 0.0);
                        light_i = 0;
                        // the draws of scene.cpp:127-163 in their order: a point on every light, the roulette, the hemisphere
                        if (v.nlights == 0) lights_done = true;
                        else {
                            const uint32_t lrec = v.off_obj + sm[v.off_lights] * OBJ_STRIDE;
                            ldir = normalize(light_point(sm, lrec, g) - hp);      // scene.cpp:129-130
                            eval_light = true;
                            if constexpr (kMulti) {
                                hn = n; hd = rd; horec = orec;
                                g_l = g;                                          // the further lights are sampled from here, later
                                for (uint32_t l = 1; l < v.nlights; ++l) light_skip(sm, v.off_obj + sm[v.off_lights + l] * OBJ_STRIDE, g);
                            }
                        }
                        p_survive = max_importance(imp);                    // scene.cpp:147, before the colour
                        impc = imp * ldd3(sm, orec + OBJ_COLOR);            // scene.cpp:148
                        survive = !((0.0 + (1.0 - 0.0) * rng_double(g)) > p_survive);   // scene.cpp:150
                        if (survive) {
                            bd = hemisphere(n, g, cosTheta);                // scene.cpp:163
                            eval_bounce = true;
                        }
                    } else if (surface == MCRT_SURFACE_SPECULAR) {
                        p = p + n * kMargin;                                // scene.cpp:171
                        rd = reflect(rd, n);
                        ro = p;
                        depth -= 1;
                    } else if (surface == MCRT_SURFACE_REFRACTIVE) {
                        const V3 I = rd;
                        float n1 = 1.0f;
                        float n2 = ldf(sm, orec + OBJ_IOR);
                        if (dot(n, I) >= 0) {                               // scene.cpp:183-187
                            n = n * -1.0f;
                            float tmp = n1; n1 = n2; n2 = tmp;
                        }
                        const V3 errorMargin = n * kMargin;
                        const float nr = n1 / n2;
                        const float cosI = dot(I, n);
                        const float cos2t = 1.0f - nr * nr * (1.0f - cosI * cosI);
                        // the reflected ray serves total internal reflection, the parked branch of a split and the single choice
                        const V3 refl_o = p + errorMargin, refl_d = reflect(I, n);
                        if (cos2t < 0.0f) {                                 // total internal reflection, scene.cpp:194-198
                            ro = refl_o; rd = refl_d; depth -= 1;
                        } else {
                            const V3 tdir = I * nr - n * (cosI * nr + sqrtf(cos2t));   // scene.cpp:202
                            float R0 = (n2 - n1) / (n2 + n1);
                            R0 *= R0;
                            const float c = 1.0f - (-cosI);
                            const float R = R0 + (1.0f - R0) * c * c * c * c * c;
                            if (rng_double(g) < max_importance(imp)) {      // scene.cpp:209-214: trace both
                                // park the reflect branch, run the refract branch first (the order g++ evaluates
                                // the two operands of the `+` at scene.cpp:212-213; pinned by fixture F5)
                                const D3 pimp = imp * (double)R;
                                pend.push(npend, refl_o, refl_d, pimp, depth - 1);
                                npend += 1;
                                if (count) rc.splits++;
                                imp = imp * (double)(1.0 - R);
                                ro = p - errorMargin; rd = tdir; depth -= 1;
                            } else {
                                const double P = .25 + .5 * R;              // scene.cpp:218-228
                                if (rng_double(g) < P) {
                                    imp = imp * (R / P);
                                    ro = refl_o; rd = refl_d; depth -= 1;
                                } else {
                                    imp = imp * ((1.0 - R) / (1.0 - P));
                                    ro = p - errorMargin; rd = tdir; depth -= 1;
                                }
                            }
                        }
                    } else {
                        branch_done = true;                                 // Diffuse_Specular: black, scene.cpp:232-233
                    }
                }
            }
        } else {   // ST_SHADOW: the query was the shadow ray towards light `light_i`, scene.cpp:133-141
            const uint32_t lobj = sm[v.off_lights + light_i];
            if (hit.obj == MISS || hit.obj == lobj) direct = direct + ldd3(sm, v.off_obj + lobj * OBJ_STRIDE + OBJ_EMISSION) * w_light;
            light_i += 1;
            if (kMulti && light_i < v.nlights) {
                const uint32_t lrec = v.off_obj + sm[v.off_lights + light_i] * OBJ_STRIDE;
                ldir = normalize(light_point(sm, lrec, g_l) - hp);
                en = hn; ed = hd; eorec = horec;
                eval_light = true;
            } else lights_done = true;
        }

        if (eval_light || eval_bounce) {
            double f_light, f_bounce;
            brdf2(sm, eorec, ed, en, eval_light, ldir, eval_bounce, bd, f_light, f_bounce);
            if (eval_light) {
                const float surfaceDot = dot(en, ldir);
                const float lightDot = dot(mk(0.0f, -1.0f, 0.0f), ldir * -1.0f);
                w_light = (double)(f_light * surfaceDot * lightDot);        // scene.cpp:136-139
                ro = hp; rd = ldir;
                state = ST_SHADOW;
            }
            if (eval_bounce) g_bounce = 2.0 * (double)cosTheta * f_bounce;  // scene.cpp:164
        }

        if (lights_done) {
            // Russian roulette and the diffuse bounce, scene.cpp:147-167 (the draws were made when the hit was shaded)
            direct = direct * impc;
            L = L + direct;
            if (!survive) branch_done = true;
            else {
                imp = ddivs(impc, p_survive);
                imp = imp * g_bounce;
                ro = hp; rd = bd; depth -= 1;
                state = ST_TRACE;
            }
        }

        if (branch_done) {
            if (npend > 0) {
                npend -= 1;
                pend.pop(npend, ro, rd, imp, depth);
                state = ST_TRACE;
            } else {
                // PixelBuffer::Accumulate, source/core/pixelbuffer.cpp:39-46 (non-finite samples dropped)
                lazy.item(*this);
                const bool fin = isfinite(L.x) && isfinite(L.y) && isfinite(L.z);
                if (fin || keep_nonfinite) sum = sum + L;
                else dropped += 1;
                if (ex) { ex->state[0] = g.s0; ex->state[1] = g.s1; }
                k += 1;
                if (k < s_count) state = ST_NEW;
                else { work.done(pixel, sum, dropped); state = ST_FETCH; }
            }
        }
    }
};

// One lane of a tracing kernel.  `work` hands out items - samples [s_begin, s_begin + s_count) of a
// pixel - one at a time (Work::next) and takes back the sum of the item's finite samples in sample
// order plus the number dropped (Work::done; keep_nonfinite: the raw value, used by the parity path
// with s_count == 1).  Every loop iteration is exactly one nearest-hit query: LaneState::pre, the query, LaneState::post.
template <class PS, class PB, int kAccel, bool kCount, bool kMulti, class Work>
__device__ __forceinline__ void trace_lane(PS sm, PB big, const SceneView& v, const CameraDev& cam, uint64_t seed, uint32_t max_depth,
                                           const PendStack& pend, bool keep_nonfinite, Counters& cn, RenderCounters& rc, Work& work,
                                           const LanePolicy policy, uint32_t* walk_lds)
{
    LaneState<kMulti> ln;
    ln.init();
    ExplicitStart exs;
    const ExplicitStart* ex = nullptr;
    const D3 background = ldd3(sm, v.off_bg);

    // Fetch policy: idle lanes wait until fetch_min of them can fetch together (or nobody is left working); they then receive
    // consecutive items - neighbouring pixels of one tile - which keeps the wavefront's rays coherent (the wave-uniform scan
    // culls whole meshes per wavefront).  fetch_min == 64 is a round: everybody waits for the slowest lane.
    const uint32_t kMinTrace = policy.min_trace, kFetchMin = policy.fetch_min;
    // Octree path: the nearest-hit query is resumable (OctreeWalk): lanes whose ray is finished are shaded
    // and given their next ray while the others stay in mid-traversal.
    constexpr bool kWalk = (kAccel == MCRT_ACCEL_OCTREE);
    OctreeWalk<PS, PB, kCount> walk;
    // walks through an HBM-resident image keep the top of their stack in LDS (walk_lds), see LdsStack
    typename WalkStackFor<PB>::type walk_stack;
    WalkStackFor<PB>::bind(walk_stack, walk_lds);
    bool walking = false;
    if (kWalk) walk.idle();
    constexpr bool kDeal = kWalk && WalkStackFor<PB>::kDeal;
    const typename OctreeWalk<PS, PB, kCount>::Deal deal = walk_deal_area<OctreeWalk<PS, PB, kCount>>(walk_lds);
    if constexpr (kDeal) { deal.slot[threadIdx.x] = 0u; wave_sync(); }      // no marker yet (tag 0 is never a deal's)

    // The loop is wave-uniform: a lane that has nothing left stays in it as a helper (the cooperative queries need all 64 lanes).
    for (;;) {
        const uint64_t busy = __ballot(ln.state != ST_DONE);
        if (busy == 0ull) break;
        bool fetch_ok = true;
        if (kFetchMin > 1u) {
            const uint64_t want = __ballot(ln.state == ST_FETCH);
            fetch_ok = (uint32_t)__popcll(want) >= kFetchMin || want == busy;
        }
#ifdef MCRT_SECTION_PROFILE
        const unsigned long long pt_pre = PROF_NOW();
        PROF_ADD(PF_ITER, 1); PROF_ADD(PF_PRE_LANES, __popcll(__ballot(ln.state == ST_NEW || (ln.state == ST_FETCH && fetch_ok))));
#endif
        // Samples start in batches too: a lane between two samples (or about to take an item) would only join the trace lanes that wait for
        // their batch, so the start - seed hash, camera ray: ~200 instructions for the wavefront, whoever needs them - runs in the iterations
        // in which that batch is served, for all the lanes that got there meanwhile, instead of in nearly every iteration for a few.
        bool start_ok = true;
        if (kMinTrace > 0u) {
            const bool starts = ln.state == ST_NEW || (ln.state == ST_FETCH && fetch_ok);
            const uint32_t n_s = (uint32_t)__popcll(__ballot(ln.state == ST_SHADOW)), n_t = (uint32_t)__popcll(__ballot(ln.state == ST_TRACE || starts));
            start_ok = n_t >= kMinTrace || n_s == 0u;
        }
        if (start_ok) ln.pre(fetch_ok, work, cam, seed, max_depth, exs, ex, rc, kCount);
#ifdef MCRT_SECTION_PROFILE
        const unsigned long long pt_cast = PROF_NOW();
        PROF_ADD(PF_PRE_T, pt_cast - pt_pre);
#endif

        Hit hit;
        {
            // Batching of the shading: a lane with a shadow ray is always served (what follows its query is one comparison and a
            // multiply-add); lanes with a camera / bounce ray are served when at least kMinTrace of them are waiting (or nobody has
            // a shadow ray), otherwise they sit this query out - after it every shadow lane has turned into a trace lane, so the
            // wait is one iteration.  The wavefront executes the whole shading of a hit whenever ANY lane needs it; this way it pays
            // for that about every other iteration, with most of its lanes in it, instead of every iteration for a minority.
            // Break-even by the instruction counts (cast ~2050, shading ~1200 per wavefront): wait when fewer than ~3/8 of the lanes are ready;
            // measured optimum of the threshold over configs 1, 2, 3 and 5: 36 (sweep 0 / 12 / 24 / 36 / 64, profiles/r02_sweep_min_trace.txt).
            bool live = ln.state == ST_TRACE || ln.state == ST_SHADOW;        // a lane waiting for an item or for its start, or finished, has no ray
            if (kMinTrace > 0u && live) {
                const bool is_s = ln.state == ST_SHADOW;
                const uint32_t n_s = (uint32_t)__popcll(__ballot(is_s)), n_t = (uint32_t)__popcll(__ballot(!is_s));
                live = is_s || n_t >= kMinTrace || n_s == 0u;
            }
            if constexpr (kWalk) {
                // a lane without a ray has an empty stack: it sits in the walk loop as a finished lane
                if constexpr (kDeal) {
                    if (live && !walking) { walk.begin_dealt(sm, big, v, walk_stack, deal, ln.ro, ln.rd, cn); walking = true; }
                    walk.run_dealt(sm, big, v, walk_stack, deal, ln.ro, ln.rd, cn, policy.refill);
                } else {
                    if (live && !walking) { walk.begin(sm, big, v, walk_stack, ln.ro, ln.rd, cn); walking = true; }
                    walk.run(sm, big, v, walk_stack, ln.ro, ln.rd, cn, policy.refill);
                }
                if (!live || !walk.done()) continue;          // no ray, or still in mid-traversal: nothing to shade yet
                walking = false;
                if constexpr (kDeal) walk.collect_dealt(sm, v, deal);
                hit = walk.best;
            } else {
                hit = cast<kAccel, PS, PB, kCount>(sm, big, v, ln.ro, ln.rd, cn, live);
#ifdef MCRT_SECTION_PROFILE
                PROF_ADD(PF_CAST_T, PROF_NOW() - pt_cast); PROF_ADD(PF_CAST_LANES, __popcll(__ballot(live)));
                PROF_ADD(PF_IDLE_FETCH, __popcll(__ballot(ln.state == ST_FETCH))); PROF_ADD(PF_IDLE_DONE, __popcll(__ballot(ln.state == ST_DONE)));
                PROF_ADD(PF_IDLE_BATCH, __popcll(__ballot(!live && ln.state != ST_FETCH && ln.state != ST_DONE)));
#endif
                if (!live) continue;
            }
        }
#ifdef MCRT_SECTION_PROFILE
        const unsigned long long pt_post = PROF_NOW();
        const uint32_t state_before = ln.state;
        const uint64_t in_post = __ballot(1);
#endif
        ln.post(hit, sm, big, v, background, pend, keep_nonfinite, work, ex, rc, kCount);
#ifdef MCRT_SECTION_PROFILE
        {
            // the `continue` above leaves only the live lanes here: lane 0 may be missing, the first lane present adds for the wavefront
            const uint64_t heavy = __ballot(state_before == ST_TRACE && ln.state == ST_SHADOW);
            const uint64_t n_t = __popcll(__ballot(state_before == ST_TRACE)), n_s = __popcll(__ballot(state_before == ST_SHADOW));
            if ((threadIdx.x & 63u) == (uint32_t)__ffsll((unsigned long long)in_post) - 1u) {
                unsigned long long* p = prof_slots();
                p[PF_POST_T] += PROF_NOW() - pt_post; p[PF_POST_TRACE_LANES] += n_t; p[PF_POST_SHADOW_LANES] += n_s;
                p[PF_POST_HEAVY] += heavy ? 1u : 0u; p[PF_POST_HEAVY_LANES] += (unsigned long long)__popcll(heavy);
            }
        }
#endif
    }
}

struct RenderArgs {
    SceneView view;
    CameraDev cam;
    double* accum;
    double* partial;            // [slices][pixels][3] when slices > 1
    uint32_t* tile_counter;     // [0] next work item
    unsigned long long* dropped;    // non-finite samples dropped since the last mcrt_clear (mcrt_dropped_samples)
    unsigned long long* counters;   // 8 x u64 (kCount builds)
    uint32_t* pend;
    void* park;                 // [chunks][launched lanes] x 16 bytes: the estimator state of the lanes that walk an HBM-resident octree (trace_lane_parked)
    uint32_t s_begin, s_count, max_depth;
    uint32_t tiles_x, tiles;    // pixel tiles of (64 >> wave_slices_log2) pixels
    uint32_t wave_slices_log2;  // slices of one pixel that share a wavefront
    uint32_t slices;            // sample slices per pixel (mcrt_lanes_per_pixel)
    uint32_t items;             // tiles * 64 * slices
    LanePolicy policy;
    uint64_t seed;
};

__device__ __forceinline__ void flush_counters(unsigned long long* out, const Counters& cn, const RenderCounters& rc)
{
    atomicAdd(out + 0, (unsigned long long)cn.rays);
    atomicAdd(out + 1, (unsigned long long)cn.nodes);
    atomicAdd(out + 2, (unsigned long long)cn.tris);
    atomicAdd(out + 3, (unsigned long long)cn.spheres);
    atomicAdd(out + 4, (unsigned long long)cn.gates);
    atomicAdd(out + 5, (unsigned long long)rc.samples);
    atomicAdd(out + 6, (unsigned long long)rc.trace_calls);
    atomicAdd(out + 7, (unsigned long long)rc.splits);
}

// Work items of a render pass: item i = (slice, 8x8 tile, pixel in tile), handed out by one atomic counter in
// tile order, so the lanes of a wavefront start on one tile and drift apart only as their paths differ in length.
// Slice l of a pixel renders samples [l*per, (l+1)*per) (the last slice takes the remainder), per = s_count / slices.
// slices == 1: the lane owns its pixel and adds its sum to the accumulator.  slices > 1 (small images, where
// pixels alone cannot fill the GPU): the slice sums go to `partial` and k_combine adds them in slice order - the
// summation order is fixed by (width, height, s_count) alone, never by scheduling.
//
// Lanes that run dry fetch together (trace_lane waits until kFetchMin of them can, Work::kFetchMin): one atomic per
// wavefront and fetch, consecutive items - pixels of one tile - for the lanes that take part.
struct RenderWork {
    const RenderArgs& a;
    uint32_t slice;
    uint32_t dropped_total;
    __device__ __forceinline__ bool decode(uint32_t i, uint32_t& pixel, uint32_t& s_begin, uint32_t& s_count)
    {
        // a wavefront's 64 items: (64 >> wave_slices_log2) pixels of one small tile x (1 << wave_slices_log2) slices
        const uint32_t t = i >> 6, within = i & 63u;
        const uint32_t tile = t % a.tiles;
        const uint32_t log_p = 6u - a.wave_slices_log2;
        const uint32_t pix = within & ((1u << log_p) - 1u);
        slice = ((t / a.tiles) << a.wave_slices_log2) + (within >> log_p);
        const uint32_t log_w = (log_p + 1u) >> 1, log_h = log_p >> 1;
        const uint32_t x = ((tile % a.tiles_x) << log_w) + (pix & ((1u << log_w) - 1u)), y = ((tile / a.tiles_x) << log_h) + (pix >> log_w);
        if (x >= a.cam.width || y >= a.cam.height) return false;
        pixel = y * a.cam.width + x;
        const uint32_t per = a.s_count / a.slices;
        s_begin = a.s_begin + slice * per;
        s_count = (slice + 1u == a.slices) ? a.s_count - slice * per : per;
        return true;
    }
    __device__ __forceinline__ bool next(uint32_t& pixel, uint32_t& s_begin, uint32_t& s_count)
    {
        for (;;) {
            // one atomic per wavefront for all lanes fetching together
            const uint64_t fetching = __ballot(1);
            const uint32_t me = threadIdx.x & 63u;
            const uint32_t leader = (uint32_t)__ffsll((unsigned long long)fetching) - 1u;
            uint32_t base = 0;
            if (me == leader) base = atomicAdd(a.tile_counter, (uint32_t)__popcll(fetching));
            base = (uint32_t)__shfl((int)base, (int)leader);
            const uint32_t i = base + (uint32_t)__popcll(fetching & ((1ull << me) - 1ull));
            if (i >= a.items) return false;
            if (decode(i, pixel, s_begin, s_count)) return true;
        }
    }
    __device__ __forceinline__ bool explicit_start(ExplicitStart&) const { return false; }
    __device__ __forceinline__ void done(uint32_t pixel, const D3& sum, uint32_t dropped)
    {
        if (a.slices == 1u) {
            double* px = a.accum + (size_t)pixel * 3;
            px[0] += sum.x; px[1] += sum.y; px[2] += sum.z;
        } else {
            double* px = a.partial + ((size_t)slice * a.cam.width * a.cam.height + pixel) * 3;
            px[0] = sum.x; px[1] = sum.y; px[2] = sum.z;
        }
        dropped_total += dropped;
    }
};

template <class PS, class PB, bool kCount, bool kMulti>
__device__ __forceinline__ void trace_lane_parked(PS sm, PB big, const SceneView& v, const CameraDev& cam, uint64_t seed, uint32_t max_depth, const PendStack& pend,
                                                  Counters& cn, RenderCounters& rc, RenderWork& work, const LanePolicy policy, uint32_t* walk_lds, U4* park, uint32_t slots);

template <bool kBigLds, int kAccel, bool kCount, bool kMulti>
__global__ __launch_bounds__(kBlock, (kAccel == MCRT_ACCEL_OCTREE && !kBigLds) ? MCRT_WALK_BLOCKS : 4) void k_render(const RenderArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    typedef SceneMem<kBigLds, kAccel> Mem;
    typename Mem::PS sm;
    typename Mem::PB big;
    Mem::open(a.view, smem, sm, big);
    uint32_t* walk_lds = smem + ((a.view.small_words + 3u) & ~3u);     // used only by LdsStack (octree, image in HBM)

    PendStack pend;
    pend.base = a.pend;
    pend.entries = a.max_depth ? a.max_depth : 1u;
    pend.lane = blockIdx.x * blockDim.x + threadIdx.x;
    Counters cn = { 0, 0, 0, 0, 0 };
    RenderCounters rc = { 0, 0, 0 };
    RenderWork work = { a, 0u, 0u };
#ifdef MCRT_SECTION_PROFILE
    if ((threadIdx.x & 63u) < 32u) prof_slots()[threadIdx.x & 63u] = 0ull;
    const unsigned long long pt_kernel = PROF_NOW();
#endif
    if constexpr (kAccel == MCRT_ACCEL_OCTREE && !kBigLds)
        trace_lane_parked<typename Mem::PS, typename Mem::PB, kCount, kMulti>(sm, big, a.view, a.cam, a.seed, a.max_depth, pend, cn, rc, work, a.policy, walk_lds, (U4*)a.park,
                                                                                gridDim.x * blockDim.x);
    else
        trace_lane<typename Mem::PS, typename Mem::PB, kAccel, kCount, kMulti>(sm, big, a.view, a.cam, a.seed, a.max_depth, pend, false, cn, rc, work, a.policy, walk_lds);
#ifdef MCRT_SECTION_PROFILE
    PROF_ADD(PF_WAVES, 1); PROF_ADD(PF_TOTAL_T, PROF_NOW() - pt_kernel);
    if ((threadIdx.x & 63u) < (uint32_t)PF_COUNT) atomicAdd(&g_prof[threadIdx.x & 63u], prof_slots()[threadIdx.x & 63u]);
#endif
    if (work.dropped_total) atomicAdd(a.dropped, (unsigned long long)work.dropped_total);
    if (kCount) flush_counters(a.counters, cn, rc);
}

// ------------------------------------------------------------------ wavefront pipeline (scenes walked from HBM)
// For a scene whose octree lives in HBM a lane of k_render spends its time in OctreeWalk, carrying the whole estimator
// state (LaneState, ~64 registers) along, at 4 wavefronts per SIMD, with lanes idle whenever their ray ends before their
// neighbours' (measured: 15-17 of 64 lanes active per instruction).  The wavefront pipeline cuts the loop at the query:
//   k_wf_shade   one thread per path SLOT whose query has been answered: LaneState::post, LaneState::pre (the same two
//                functions k_render runs), state back to HBM, the slot appended to the queue of slots with a new ray;
//   k_wf_trace   persistent wavefronts that only walk: a lane takes the next queued slot the moment its ray is finished
//                (one aggregated atomic per wavefront and fetch), so every lane walks all the time, and the kernel needs
//                half the registers: more wavefronts per SIMD to hide the dependent loads of a visit.
// A slot is a virtual lane of k_render: it works off its items (sample slices of pixels) one sample after the other and
// hands back sums in sample order, so the accumulators are bit-identical to k_render's.  State lives in HBM as 16-byte
// chunks, chunk-major (chunk c of slot s at [c * slots + s]): consecutive slots, coalesced accesses.
constexpr uint32_t kWfChunks = 16, kWfChunksMulti = 19;
constexpr uint32_t WF_NONE = 0xFFFFFFFFu;
constexpr uint32_t kWfBatch = 64;           // queue entries a wavefront of k_wf_trace takes per atomic
enum : uint32_t { WF_CNT_Q0 = 0, WF_CNT_Q1 = 1, WF_CNT_CURSOR = 2, WF_CNT_WORDS = 8 };

struct WfArgs {
    RenderArgs r;
    U4* state;               // [chunks][slots]
    U4* hits;                // [slots]: prim, obj, elem, t of the slot's last query
    const uint32_t* q_in;    // slots whose query was answered (shade) / slots with a ray (trace)
    uint32_t* q_out;         // shade: slots that have a new ray
    uint32_t* cnt;           // WF_CNT_*
    uint32_t in_idx, out_idx;   // which of cnt[Q0], cnt[Q1] counts q_in / q_out
    uint32_t slots;
    uint32_t first;          // shade: no answered queries yet - every slot starts from scratch
};


template <bool kMulti>
__device__ __forceinline__ void wf_store(const LaneState<kMulti>& s, uint32_t slice, U4* st, uint32_t slots, uint32_t slot)
{
    U4* p = st + slot;
    const size_t n = slots;
    p[0 * n] = u4(__float_as_uint(s.ro.x), __float_as_uint(s.ro.y), __float_as_uint(s.ro.z), s.state);
    p[1 * n] = u4(__float_as_uint(s.rd.x), __float_as_uint(s.rd.y), __float_as_uint(s.rd.z), s.depth);
    p[2 * n] = u4((uint32_t)s.g.s0, (uint32_t)(s.g.s0 >> 32), (uint32_t)s.g.s1, (uint32_t)(s.g.s1 >> 32));
    p[3 * n] = dd4(s.imp.x, s.imp.y);
    p[4 * n] = dd4(s.imp.z, s.L.x);
    p[5 * n] = dd4(s.L.y, s.L.z);
    p[6 * n] = dd4(s.sum.x, s.sum.y);
    p[7 * n] = u4(dlo(s.sum.z), dhi(s.sum.z), s.pixel, s.s_begin);
    p[8 * n] = u4(s.s_count, s.k, s.dropped, s.npend);
    p[9 * n] = u4(__float_as_uint(s.hp.x), __float_as_uint(s.hp.y), __float_as_uint(s.hp.z), s.light_i | (s.survive ? 0x80000000u : 0u));
    p[10 * n] = u4(__float_as_uint(s.bd.x), __float_as_uint(s.bd.y), __float_as_uint(s.bd.z), slice);
    p[11 * n] = dd4(s.impc.x, s.impc.y);
    p[12 * n] = dd4(s.impc.z, s.direct.x);
    p[13 * n] = dd4(s.direct.y, s.direct.z);
    p[14 * n] = dd4(s.p_survive, s.g_bounce);
    p[15 * n] = u4(dlo(s.w_light), dhi(s.w_light), s.horec, 0u);
    if constexpr (kMulti) {
        p[16 * n] = u4(__float_as_uint(s.hn.x), __float_as_uint(s.hn.y), __float_as_uint(s.hn.z), 0u);
        p[17 * n] = u4(__float_as_uint(s.hd.x), __float_as_uint(s.hd.y), __float_as_uint(s.hd.z), 0u);
        p[18 * n] = u4((uint32_t)s.g_l.s0, (uint32_t)(s.g_l.s0 >> 32), (uint32_t)s.g_l.s1, (uint32_t)(s.g_l.s1 >> 32));
    }
}

template <bool kMulti>
__device__ __forceinline__ void wf_load(LaneState<kMulti>& s, uint32_t& slice, const U4* st, uint32_t slots, uint32_t slot)
{
    const U4* p = st + slot;
    const size_t n = slots;
    U4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = p[(size_t)i * n];
    s.ro = mk(__uint_as_float(c[0].x), __uint_as_float(c[0].y), __uint_as_float(c[0].z)); s.state = c[0].w;
    s.rd = mk(__uint_as_float(c[1].x), __uint_as_float(c[1].y), __uint_as_float(c[1].z)); s.depth = c[1].w;
    s.g.s0 = (uint64_t)c[2].x | ((uint64_t)c[2].y << 32); s.g.s1 = (uint64_t)c[2].z | ((uint64_t)c[2].w << 32);
    s.imp = dmk(dmake(c[3].x, c[3].y), dmake(c[3].z, c[3].w), dmake(c[4].x, c[4].y));
    s.L = dmk(dmake(c[4].z, c[4].w), dmake(c[5].x, c[5].y), dmake(c[5].z, c[5].w));
    s.sum = dmk(dmake(c[6].x, c[6].y), dmake(c[6].z, c[6].w), dmake(c[7].x, c[7].y));
    s.pixel = c[7].z; s.s_begin = c[7].w;
    s.s_count = c[8].x; s.k = c[8].y; s.dropped = c[8].z; s.npend = c[8].w;
    s.hp = mk(__uint_as_float(c[9].x), __uint_as_float(c[9].y), __uint_as_float(c[9].z));
    s.light_i = c[9].w & 0x7FFFFFFFu; s.survive = (c[9].w >> 31) != 0u;
    s.bd = mk(__uint_as_float(c[10].x), __uint_as_float(c[10].y), __uint_as_float(c[10].z)); slice = c[10].w;
    s.impc = dmk(dmake(c[11].x, c[11].y), dmake(c[11].z, c[11].w), dmake(c[12].x, c[12].y));
    s.direct = dmk(dmake(c[12].z, c[12].w), dmake(c[13].x, c[13].y), dmake(c[13].z, c[13].w));
    s.p_survive = dmake(c[14].x, c[14].y); s.g_bounce = dmake(c[14].z, c[14].w);
    s.w_light = dmake(c[15].x, c[15].y); s.horec = c[15].z;
    if constexpr (kMulti) {
        const U4 a = p[16 * n], b = p[17 * n], e = p[18 * n];
        s.hn = mk(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        s.hd = mk(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
        s.g_l.s0 = (uint64_t)e.x | ((uint64_t)e.y << 32); s.g_l.s1 = (uint64_t)e.z | ((uint64_t)e.w << 32);
    } else {
        s.hn = mk(0, 0, 0); s.hd = mk(0, 0, 0); s.g_l.s0 = 0; s.g_l.s1 = 0;
    }
}

// The render loop for an octree walked from HBM (the 1M-triangle mesh).  trace_lane keeps the whole estimator state (LaneState, ~64
// registers) live across the walk; at the 128 registers of four wavefronts per SIMD the compiler then spills the walk's own hot
// variables (the stack pointer, the list counter) to scratch, and every reload waits for ALL outstanding loads - the node and leaf
// records included, which serialises the very round trips the walk tries to overlap.  Here the state is PARKED in HBM between two
// shadings (ParkIO: 16-byte chunks in three groups, lane-major inside a group): a lane carries its ray and
// its state word through the walk and nothing else, and loads the rest only when its query is answered - post, pre, store.  Same
// LaneState code, same numbers.  About 256 bytes of coalesced traffic per query against the ~3 KB of records a walk fetches.
// The parked state in three groups, so that a shading pass moves what it needs (16-byte chunks; group by group, lane-major inside a group):
//   A  chunks 0-3    generator, importance, radiance: every pass, both ways
//   I  chunks 4-6    the item (sum, pixel, sample range, counters, slice): read and written back by the lanes whose sample ends (LaneState::post
//                    asks for it through `lazy.item`), i.e. once per sample instead of once per query
//   D  chunks 7-13   the diffuse-hit context (+ 14-16 with several lights): written by a lane that leaves with a shadow ray, read by a lane that
//                    comes back from one - every other pass of a diffuse path, one way each
// The ray, the state word, the remaining depth, the parked-branch count and the roulette outcome never leave the registers
// (state | depth << 3 | npend << 12 | survive << 30).
// 8 chunks per lane and pass on average instead of 32.
constexpr uint32_t kParkChunksMulti = 20;     // 17 used with several lights, 14 with one (lane-major groups are padded to whole sectors)
template <bool kMulti> struct ParkIO {
    U4* base;       // the parked state of the launch
    size_t n;       // launched lanes
    size_t slot;    // this lane
    // lane-major inside a group: the chunks of one lane's group fill whole 64-byte sectors of their own (A: 64 B, I: 48 of 64 B, D: 64 B,
    // 160 of 192 B with several lights), so a pass in which some of the lanes take part moves those lanes' sectors and no others
    // (chunk-major columns, four lanes to a sector, moved 1.5x the bytes: 46.2 -> 45.4 ms)
    static constexpr size_t kD = kMulti ? 12 : 4;
    __device__ __forceinline__ U4& at(uint32_t k) const
    {
        if (k < 4u) return base[slot * 4u + k];
        if (k < 7u) return base[n * 4u + slot * 4u + (k - 4u)];
        return base[n * 8u + slot * kD + (k - 7u)];
    }
    __device__ __forceinline__ void load_a(LaneState<kMulti>& s) const
    {
        const U4 c0 = at(0), c1 = at(1), c2 = at(2), c3 = at(3);
        s.g.s0 = (uint64_t)c0.x | ((uint64_t)c0.y << 32); s.g.s1 = (uint64_t)c0.z | ((uint64_t)c0.w << 32);
        s.imp = dmk(dmake(c1.x, c1.y), dmake(c1.z, c1.w), dmake(c2.x, c2.y));
        s.L = dmk(dmake(c2.z, c2.w), dmake(c3.x, c3.y), dmake(c3.z, c3.w));
    }
    __device__ __forceinline__ void store_a(const LaneState<kMulti>& s) const
    {
        at(0) = u4((uint32_t)s.g.s0, (uint32_t)(s.g.s0 >> 32), (uint32_t)s.g.s1, (uint32_t)(s.g.s1 >> 32));
        at(1) = dd4(s.imp.x, s.imp.y);
        at(2) = dd4(s.imp.z, s.L.x);
        at(3) = dd4(s.L.y, s.L.z);
    }
    __device__ __forceinline__ void load_i(LaneState<kMulti>& s, uint32_t& slice) const
    {
        const U4 c4 = at(4), c5 = at(5), c6 = at(6);
        s.sum = dmk(dmake(c4.x, c4.y), dmake(c4.z, c4.w), dmake(c5.x, c5.y));
        s.pixel = c5.z; s.s_begin = c5.w;
        s.s_count = c6.x; s.k = c6.y; s.dropped = c6.z; slice = c6.w;
    }
    __device__ __forceinline__ void store_i(const LaneState<kMulti>& s, uint32_t slice) const
    {
        at(4) = dd4(s.sum.x, s.sum.y);
        at(5) = u4(dlo(s.sum.z), dhi(s.sum.z), s.pixel, s.s_begin);
        at(6) = u4(s.s_count, s.k, s.dropped, slice);
    }
    __device__ __forceinline__ void load_d(LaneState<kMulti>& s) const
    {
        if constexpr (!kMulti) {
            // one light: 64 bytes, one sector - the light index and the direct sum are zero while the shadow ray of the only light is out,
            // the roulette threshold is a function of the importance (group A, unchanged since the hit), `survive` travels in the state word
            const U4 c7 = at(7), c8 = at(8), c9 = at(9), c10 = at(10);
            s.hp = mk(__uint_as_float(c7.x), __uint_as_float(c7.y), __uint_as_float(c7.z));
            s.bd = mk(__uint_as_float(c7.w), __uint_as_float(c8.x), __uint_as_float(c8.y));
            s.g_bounce = dmake(c8.z, c8.w);
            s.impc = dmk(dmake(c9.x, c9.y), dmake(c9.z, c9.w), dmake(c10.x, c10.y));
            s.w_light = dmake(c10.z, c10.w);
            s.light_i = 0u; s.direct = dmk(0.0, 0.0, 0.0);
            s.p_survive = max_importance(s.imp);
            return;
        }
        const U4 c7 = at(7), c8 = at(8), c9 = at(9), c10 = at(10), c11 = at(11), c12 = at(12), c13 = at(13);
        s.hp = mk(__uint_as_float(c7.x), __uint_as_float(c7.y), __uint_as_float(c7.z));
        s.light_i = c7.w & 0x7FFFFFFFu; s.survive = (c7.w >> 31) != 0u;
        s.bd = mk(__uint_as_float(c8.x), __uint_as_float(c8.y), __uint_as_float(c8.z)); s.horec = c8.w;
        s.impc = dmk(dmake(c9.x, c9.y), dmake(c9.z, c9.w), dmake(c10.x, c10.y));
        s.direct = dmk(dmake(c10.z, c10.w), dmake(c11.x, c11.y), dmake(c11.z, c11.w));
        s.p_survive = dmake(c12.x, c12.y); s.g_bounce = dmake(c12.z, c12.w);
        s.w_light = dmake(c13.x, c13.y);
        if constexpr (kMulti) {
            const U4 a = at(14), b = at(15), e = at(16);
            s.hn = mk(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
            s.hd = mk(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
            s.g_l.s0 = (uint64_t)e.x | ((uint64_t)e.y << 32); s.g_l.s1 = (uint64_t)e.z | ((uint64_t)e.w << 32);
        }
    }
    __device__ __forceinline__ void store_d(const LaneState<kMulti>& s) const
    {
        if constexpr (!kMulti) {
            at(7) = u4(__float_as_uint(s.hp.x), __float_as_uint(s.hp.y), __float_as_uint(s.hp.z), __float_as_uint(s.bd.x));
            at(8) = u4(__float_as_uint(s.bd.y), __float_as_uint(s.bd.z), dlo(s.g_bounce), dhi(s.g_bounce));
            at(9) = dd4(s.impc.x, s.impc.y);
            at(10) = dd4(s.impc.z, s.w_light);
            return;
        }
        at(7) = u4(__float_as_uint(s.hp.x), __float_as_uint(s.hp.y), __float_as_uint(s.hp.z), s.light_i | (s.survive ? 0x80000000u : 0u));
        at(8) = u4(__float_as_uint(s.bd.x), __float_as_uint(s.bd.y), __float_as_uint(s.bd.z), s.horec);
        at(9) = dd4(s.impc.x, s.impc.y);
        at(10) = dd4(s.impc.z, s.direct.x);
        at(11) = dd4(s.direct.y, s.direct.z);
        at(12) = dd4(s.p_survive, s.g_bounce);
        at(13) = u4(dlo(s.w_light), dhi(s.w_light), 0u, 0u);
        if constexpr (kMulti) {
            at(14) = u4(__float_as_uint(s.hn.x), __float_as_uint(s.hn.y), __float_as_uint(s.hn.z), 0u);
            at(15) = u4(__float_as_uint(s.hd.x), __float_as_uint(s.hd.y), __float_as_uint(s.hd.z), 0u);
            at(16) = u4((uint32_t)s.g_l.s0, (uint32_t)(s.g_l.s0 >> 32), (uint32_t)s.g_l.s1, (uint32_t)(s.g_l.s1 >> 32));
        }
    }
};
// LaneState::post's hook: the item's fields arrive when the lane's sample ends
template <bool kMulti> struct ParkLazy {
    const ParkIO<kMulti>& io;
    RenderWork& work;
    bool touched;
    __device__ __forceinline__ void item(LaneState<kMulti>& s)
    {
        if (!touched) { io.load_i(s, work.slice); touched = true; }
    }
};
__device__ __forceinline__ uint32_t park_pack(uint32_t state, uint32_t depth, uint32_t npend, bool survive) { return state | (depth << 3) | (npend << 12) | (survive ? 0x40000000u : 0u); }

template <class PS, class PB, bool kCount, bool kMulti>
__device__ __forceinline__ void trace_lane_parked(PS sm, PB big, const SceneView& v, const CameraDev& cam, uint64_t seed, uint32_t max_depth, const PendStack& pend,
                                                  Counters& cn, RenderCounters& rc, RenderWork& work, const LanePolicy policy, uint32_t* walk_lds, U4* park, uint32_t slots)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const ParkIO<kMulti> io = { park, (size_t)slots, (size_t)slot };
    uint32_t sd = park_pack(ST_FETCH, 0u, 0u, false);      // state | depth << 3 | npend << 12 | survive << 30
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    {
        LaneState<kMulti> ln;
        ln.init();
        io.store_a(ln); io.store_i(ln, 0u);
    }
    ExplicitStart exs;
    const ExplicitStart* ex = nullptr;
    const uint32_t kMinTrace = policy.min_trace, kFetchMin = policy.fetch_min;
    typedef OctreeWalk<PS, PB, kCount> Walk;
    Walk walk;
    typename WalkStackFor<PB>::type walk_stack;
    WalkStackFor<PB>::bind(walk_stack, walk_lds);
    bool walking = false;
    walk.idle();
    const typename Walk::Deal deal = walk_deal_area<Walk>(walk_lds);
    deal.slot[threadIdx.x] = 0u;          // no marker yet (tag 0 is never a deal's)
    wave_sync();
#ifdef MCRT_SECTION_PROFILE
    unsigned long long pt_shade = 0ull;
#endif
    for (;;) {
#ifdef MCRT_SECTION_PROFILE
        const unsigned long long pt_top = PROF_NOW();
        if (pt_shade) PROF_ADD(PF_POST_T, pt_top - pt_shade);
        PROF_ADD(PF_ITER, 1);
#endif
        const uint32_t state = sd & 7u;
        const uint64_t busy = __ballot(state != ST_DONE);
        if (busy == 0ull) break;
        bool fetch_ok = true;
        if (kFetchMin > 1u) {
            const uint64_t want = __ballot(state == ST_FETCH);
            fetch_ok = (uint32_t)__popcll(want) >= kFetchMin || want == busy;
        }
        if ((state == ST_FETCH && fetch_ok) || state == ST_NEW) {
            // (the start of the launch, and lanes that waited to fetch together: the diffuse-hit context is dead in both states)
            LaneState<kMulti> ln;
            ln.init();
            ln.state = state; ln.depth = (sd >> 3) & 0x1FFu; ln.npend = (sd >> 12) & 0x1FFu;
            io.load_a(ln); io.load_i(ln, work.slice);
            ln.pre(fetch_ok, work, cam, seed, max_depth, exs, ex, rc, kCount);
            io.store_a(ln); io.store_i(ln, work.slice);
            sd = park_pack(ln.state, ln.depth, ln.npend, ln.survive); ro = ln.ro; rd = ln.rd;
        }
        const uint32_t st = sd & 7u;
        bool live = st != ST_FETCH && st != ST_DONE;
        if (kMinTrace > 0u && live) {
            const bool is_s = st == ST_SHADOW;
            const uint32_t n_s = (uint32_t)__popcll(__ballot(is_s)), n_t = (uint32_t)__popcll(__ballot(!is_s));
            live = is_s || n_t >= kMinTrace || n_s == 0u;
        }
#ifdef MCRT_SECTION_PROFILE
        const unsigned long long pt_walk = PROF_NOW();
        PROF_ADD(PF_PRE_T, pt_walk - pt_top);
        PROF_ADD(PF_PRE_LANES, __popcll(__ballot(live && !walking)));
#endif
        if (live && !walking) { walk.begin_dealt(sm, big, v, walk_stack, deal, ro, rd, cn); walking = true; }
        walk.run_dealt(sm, big, v, walk_stack, deal, ro, rd, cn, policy.refill);
#ifdef MCRT_SECTION_PROFILE
        pt_shade = PROF_NOW();
        PROF_ADD(PF_CAST_T, pt_shade - pt_walk);
        PROF_ADD(PF_CAST_LANES, __popcll(__ballot(live)));
        PROF_ADD(PF_POST_TRACE_LANES, __popcll(__ballot(live && walk.done())));
#endif
        if (!live || !walk.done()) continue;          // no ray, or still in mid-traversal: nothing to shade yet
        walking = false;
        {
            LaneState<kMulti> ln;
            ln.init();
            ln.state = st; ln.depth = (sd >> 3) & 0x1FFu; ln.npend = (sd >> 12) & 0x1FFu; ln.survive = (sd >> 30) != 0u; ln.ro = ro; ln.rd = rd;
            io.load_a(ln);
            if (st == ST_SHADOW) io.load_d(ln);
            walk.collect_dealt(sm, v, deal);          // (LDS reads: they run under the HBM round trip of the loads above)
            ParkLazy<kMulti> lazy = { io, work, false };
            const D3 background = ldd3(sm, v.off_bg);
            ln.post(walk.best, sm, big, v, background, pend, false, work, ex, rc, kCount, lazy);
            if (ln.state == ST_NEW || (ln.state == ST_FETCH && kFetchMin <= 1u)) ln.pre(true, work, cam, seed, max_depth, exs, ex, rc, kCount);
            io.store_a(ln);
            if (ln.state == ST_SHADOW) io.store_d(ln);
            if (lazy.touched) io.store_i(ln, work.slice);
            sd = park_pack(ln.state, ln.depth, ln.npend, ln.survive); ro = ln.ro; rd = ln.rd;
        }
    }
}

// One step of every slot whose query has been answered: shade the hit, get the next ray ready (next sample / next item included).
template <bool kCount, bool kMulti>
__global__ __launch_bounds__(kBlock, 4) void k_wf_shade(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    typedef SceneMem<false, MCRT_ACCEL_OCTREE> Mem;
    typename Mem::PS sm;
    typename Mem::PB big;
    Mem::open(a.r.view, smem, sm, big);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.cnt[WF_CNT_CURSOR] = 0;     // the trace kernel that follows starts at the head of q_out
    const D3 background = ldd3(sm, a.r.view.off_bg);
    Counters cn = { 0, 0, 0, 0, 0 };
    RenderCounters rc = { 0, 0, 0 };
    RenderWork work = { a.r, 0u, 0u };
    PendStack pend;
    pend.base = a.r.pend;
    pend.entries = a.r.max_depth ? a.r.max_depth : 1u;
    ExplicitStart exs;
    const ExplicitStart* ex = nullptr;

    const uint32_t n = a.first ? a.slots : a.cnt[a.in_idx];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stride = gridDim.x * blockDim.x;
    // whole wavefronts go round together (the ballots of the queue append need them)
    for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + lane;
        const bool have = i < n;
        uint32_t slot = WF_NONE;
        LaneState<kMulti> ln;
        bool active = false;
        if (have) {
            slot = a.first ? i : a.q_in[i];
            pend.lane = slot;
            if (a.first) { ln.init(); work.slice = 0; }
            else {
                wf_load(ln, work.slice, a.state, a.slots, slot);
                const U4 h = a.hits[slot];
                Hit hit; hit.prim = h.x; hit.obj = h.y; hit.elem = h.z; hit.t = __uint_as_float(h.w);
                ln.post(hit, sm, big, a.r.view, background, pend, false, work, ex, rc, kCount);
            }
            ln.pre(true, work, a.r.cam, a.r.seed, a.r.max_depth, exs, ex, rc, kCount);
            active = ln.state != ST_DONE;
            if (active) wf_store(ln, work.slice, a.state, a.slots, slot);
        }
        // append the slots that have a ray to q_out: one atomic per wavefront
        const uint64_t m = __ballot(active);
        if (m) {
            const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
            uint32_t at = 0;
            if (lane == leader) at = atomicAdd(a.cnt + a.out_idx, (uint32_t)__popcll(m));
            at = (uint32_t)__shfl((int)at, (int)leader);
            if (active) a.q_out[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
        }
    }
    if (work.dropped_total) atomicAdd(a.r.dropped, (unsigned long long)work.dropped_total);
    if (kCount) flush_counters(a.r.counters, cn, rc);
}

// Persistent wavefronts that walk the octree for the queued slots; a lane takes the next slot the moment its ray is finished.
template <bool kCount, int kStackEntries, int kBlocksPerCu>
__global__ __launch_bounds__(kBlock, kBlocksPerCu) void k_wf_trace(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const SceneView& v = a.r.view;
    stage_scene<false>(v, smem);
    const LdsPtr sm = (LdsPtr)smem;
    const GlbPtr big = v.big_img;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.cnt[a.out_idx] = 0;        // the shade kernel that follows appends to an empty queue
    LdsStackT<kStackEntries> st;
    st.col = (__attribute__((address_space(3))) uint32_t*)(smem + ((v.small_words + 3u) & ~3u)) + threadIdx.x;
    Counters cn = { 0, 0, 0, 0, 0 };
    RenderCounters rc = { 0, 0, 0 };
    OctreeWalk<LdsPtr, GlbPtr, kCount> walk;
    walk.idle();
    const uint32_t n = a.cnt[a.in_idx];
    const uint32_t lane = threadIdx.x & 63u;
    const GlbPtr nodes = big + v.off_node, leaves = big + v.off_leaf;
    uint32_t slot = WF_NONE;
    bool exhausted = false;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), inv = mk(0, 0, 0), padt = mk(0, 0, 0), dts = mk(0, 0, 0);
    uint32_t flip = 0;
    // The wavefront owns a batch of queue entries [w_pos, w_end) (wave-uniform): lanes that need a ray take from it, the
    // batch is refilled with ONE atomic per kWfBatch entries - a per-fetch atomic on the one cursor serialised in L2
    // (measured: 2 M rays took 12 ms of which the atomics were most).
    uint32_t w_pos = 0, w_end = 0;
    for (;;) {
        const bool idle_lane = walk.done();
        if (idle_lane && slot != WF_NONE) {
            a.hits[slot] = u4(walk.best.prim, walk.best.obj, walk.best.elem, __float_as_uint(walk.best.t));
            slot = WF_NONE;
        }
        uint64_t need = __ballot(idle_lane && !exhausted);
        while (need) {
            if (w_pos >= w_end) {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(a.cnt + WF_CNT_CURSOR, kWfBatch);
                at = (uint32_t)__shfl((int)at, 0);
                w_pos = at < n ? at : n;
                w_end = (at + kWfBatch) < n ? (at + kWfBatch) : n;
                if (w_pos >= w_end) { exhausted = true; break; }       // the queue is empty: wave-uniform
            }
            const uint32_t avail = w_end - w_pos;
            const bool mine = (need >> lane) & 1ull;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const bool take = mine && rank < avail;
            if (take) {
                slot = a.q_in[w_pos + rank];
                const U4 c0 = a.state[slot], c1 = a.state[(size_t)a.slots + slot];
                o = mk(__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z));
                d = mk(__uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z));
                OctreeWalk<LdsPtr, GlbPtr, kCount>::ray_setup(v, d, inv, padt, dts, flip);
                walk.begin(sm, big, v, st, o, d, cn);
            }
            const uint32_t taken = (uint32_t)__popcll(need) < avail ? (uint32_t)__popcll(need) : avail;
            w_pos += taken;
            need = __ballot(mine && !take);
        }
        if (__ballot(!walk.done()) == 0ull) break;     // nobody walks: every lane has stored its hit and found the queue empty
        if (!walk.done()) walk.visit(sm, nodes, leaves, v, st, o, d, inv, padt, dts, flip, cn);
    }
    if (kCount) flush_counters(a.r.counters, cn, rc);
}

// accumulator += ((slice 0 + slice 1) + slice 2) + ... per pixel and channel: the order the test oracle uses
__global__ __launch_bounds__(kBlock) void k_combine(double* __restrict__ accum, const double* __restrict__ partial, uint64_t values, uint32_t slices)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < values; i += stride) {
        double tot = partial[i];
        for (uint32_t l = 1; l < slices; ++l) tot += partial[(uint64_t)l * values + i];
        accum[i] += tot;
    }
}

// accumulator += another accumulator of the same size (the contexts of one GPU, or a peer buffer): mcrt_add_accumulator
__global__ __launch_bounds__(kBlock) void k_add(double* __restrict__ accum, const double* __restrict__ src, uint64_t values)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < values; i += stride) accum[i] += src[i];
}

struct ListArgs {
    SceneView view;
    CameraDev cam;
    const uint32_t* pixels;
    const uint32_t* samples;
    double* out;
    unsigned long long* counters;
    uint32_t* pend;
    uint64_t n;
    uint32_t max_depth;
    uint64_t seed;
    // explicit rays (mcrt_trace_rays) when ray_o != null
    const float* ray_o;
    const float* ray_d;
    uint64_t* ray_state;
    const double* ray_importance;
    uint32_t refill;
};

// Work items of mcrt_trace_samples / mcrt_trace_rays: entry i of the caller's arrays, grid-strided per lane.
struct ListWork {
    const ListArgs& a;
    uint64_t i, stride, cur;
    __device__ __forceinline__ bool next(uint32_t& pixel, uint32_t& s_begin, uint32_t& s_count)
    {
        if (i >= a.n) return false;
        cur = i; i += stride;
        pixel = a.ray_o ? 0u : a.pixels[cur];
        s_begin = a.ray_o ? 0u : a.samples[cur];
        s_count = 1u;
        return true;
    }
    __device__ __forceinline__ bool explicit_start(ExplicitStart& e) const
    {
        if (!a.ray_o) return false;
        e.o = a.ray_o + 3 * cur; e.d = a.ray_d + 3 * cur; e.state = a.ray_state + 2 * cur;
        e.importance = a.ray_importance ? a.ray_importance + 3 * cur : nullptr;
        return true;
    }
    __device__ __forceinline__ void done(uint32_t, const D3& sum, uint32_t)
    {
        a.out[3 * cur] = sum.x; a.out[3 * cur + 1] = sum.y; a.out[3 * cur + 2] = sum.z;
    }
};

template <bool kBigLds, int kAccel, bool kCount, bool kMulti>
__global__ __launch_bounds__(kBlock) void k_trace_list(const ListArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    typedef SceneMem<kBigLds, kAccel> Mem;
    typename Mem::PS sm;
    typename Mem::PB big;
    Mem::open(a.view, smem, sm, big);
    uint32_t* walk_lds = smem + ((a.view.small_words + 3u) & ~3u);     // used only by LdsStack (octree, image in HBM)
    PendStack pend;
    pend.base = a.pend;
    pend.entries = a.max_depth ? a.max_depth : 1u;
    pend.lane = blockIdx.x * blockDim.x + threadIdx.x;
    Counters cn = { 0, 0, 0, 0, 0 };
    RenderCounters rc = { 0, 0, 0 };
    ListWork work = { a, pend.lane, (uint64_t)gridDim.x * blockDim.x, 0 };
    const LanePolicy policy = { 0u, 1u, a.refill };
    trace_lane<typename Mem::PS, typename Mem::PB, kAccel, kCount, kMulti>(sm, big, a.view, a.cam, a.seed, a.max_depth, pend, true, cn, rc, work, policy, walk_lds);
    if (kCount) flush_counters(a.counters, cn, rc);
}

template <bool kBigLds, int kAccel, bool kCount>
__global__ __launch_bounds__(kBlock) void k_intersect(const SceneView v, uint64_t n, const float* __restrict__ o, const float* __restrict__ d,
                                                      int32_t* __restrict__ out_obj, uint32_t* __restrict__ out_elem, float* __restrict__ out_t,
                                                      unsigned long long* counters)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    typedef SceneMem<kBigLds, kAccel> Mem;
    typename Mem::PS sm;
    typename Mem::PB big;
    Mem::open(v, smem, sm, big);
    Counters cn = { 0, 0, 0, 0, 0 };
    RenderCounters rc = { 0, 0, 0 };
    // every wavefront runs the same number of trips with all 64 lanes (the cooperative queries need them): lanes past the end help
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint64_t i = base + (threadIdx.x & 63u);
        const bool live = i < n;
        const uint64_t k = live ? i : 0;
        const V3 ro = mk(o[3 * k], o[3 * k + 1], o[3 * k + 2]);
        const V3 rd = mk(d[3 * k], d[3 * k + 1], d[3 * k + 2]);
        const Hit h = cast<kAccel, typename Mem::PS, typename Mem::PB, kCount>(sm, big, v, ro, rd, cn, live);
        if (live) {
            out_obj[i] = (h.obj == MISS) ? -1 : (int32_t)h.obj;
            out_elem[i] = h.elem;
            out_t[i] = h.t;
        }
    }
    if (kCount) flush_counters(counters, cn, rc);
}

__global__ __launch_bounds__(kBlock) void k_camera_rays(const CameraDev cam, uint64_t n, const float* __restrict__ xs, const float* __restrict__ ys,
                                                        float* __restrict__ out_o, float* __restrict__ out_d)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V3 o, d;
        camera_ray(cam, xs[i], ys[i], o, d);
        out_o[3 * i] = o.x; out_o[3 * i + 1] = o.y; out_o[3 * i + 2] = o.z;
        out_d[3 * i] = d.x; out_d[3 * i + 1] = d.y; out_d[3 * i + 2] = d.z;
    }
}

template <bool kBigLds, int kAccel>
__global__ __launch_bounds__(kBlock) void k_unlit(const SceneView v, const CameraDev cam, double* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    typedef SceneMem<kBigLds, kAccel> Mem;
    typename Mem::PS sm;
    typename Mem::PB big;
    Mem::open(v, smem, sm, big);
    Counters cn = { 0, 0, 0, 0, 0 };
    const uint64_t npix = (uint64_t)cam.width * cam.height;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < npix; base += stride) {
        const uint64_t i = base + (threadIdx.x & 63u);
        const bool live = i < npix;
        const uint64_t k = live ? i : 0;
        const uint32_t x = (uint32_t)(k % cam.width), y = (uint32_t)(k / cam.width);
        V3 o, d;
        camera_ray(cam, (float)x + 0.5f, (float)y + 0.5f, o, d);     // source/main.cpp:126
        const Hit h = cast<kAccel, typename Mem::PS, typename Mem::PB, false>(sm, big, v, o, d, cn, live);
        if (live) {
            D3 c = dmk(0.0, 0.0, 0.0);
            if (h.obj != MISS) c = ldd3(sm, v.off_obj + h.obj * OBJ_STRIDE + OBJ_COLOR);
            out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
        }
    }
}

// ---- the per-object virtuals of class Object (source/objects/object.h:23-29) on ray / point batches: what the host mirror's
// Object::Intersects, GetSurfaceNormal, GetRandomPointOnSurface and Material::BRDF forward to.  SMALL is staged in LDS, BIG read in place.

// Object::Intersects of ONE object: SphereObject::Intersects (source/objects/sphere.h:16-52) or TriangleMesh::Intersects
// (source/objects/mesh.cpp:9-41: the AABB gate, then every triangle, the first of the nearest kept).
__global__ __launch_bounds__(kBlock) void k_object_intersect(const SceneView v, uint32_t obj, uint64_t n, const float* __restrict__ o, const float* __restrict__ d,
                                                             uint8_t* __restrict__ out_hit, uint32_t* __restrict__ out_elem, float* __restrict__ out_t)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    stage_scene<false>(v, smem);
    const LdsPtr sm = (LdsPtr)smem;
    const GlbPtr big = v.big_img;
    const uint32_t orec = v.off_obj + obj * OBJ_STRIDE;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const V3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        bool hit = false;
        uint32_t elem = 0;
        float t = kFltMax;
        if (sm[orec + OBJ_KIND] == (uint32_t)MCRT_OBJECT_SPHERE) {
            float ts;
            if (sphere_hit(sm, orec, ro, rd, ts)) { hit = true; t = ts; }
        } else if (gate_hit(sm, orec, ro, rd)) {
            const uint32_t first = sm[orec + OBJ_FIRST_TRI], count = sm[orec + OBJ_TRI_COUNT];
            float nearest = kFltMax, hitDistance = kFltMax;
            for (uint32_t k = 0; k < count; ++k)
                if (tri_hit(big, v.off_tri + (first + k) * TRI_STRIDE, ro, rd, hitDistance) && hitDistance < nearest) { nearest = hitDistance; elem = k; }
            if (nearest < kFltMax) { hit = true; t = nearest; } else elem = 0;
        }
        out_hit[i] = hit ? 1 : 0; out_elem[i] = elem; out_t[i] = t;
    }
}

// Object::GetSurfaceNormal(location, index): source/objects/sphere.h:54-57, source/objects/mesh.cpp:43-46
__global__ __launch_bounds__(kBlock) void k_surface_normals(const SceneView v, uint64_t n, const uint32_t* __restrict__ objs, const uint32_t* __restrict__ elems,
                                                            const float* __restrict__ pos, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    stage_scene<false>(v, smem);
    const LdsPtr sm = (LdsPtr)smem;
    const GlbPtr big = v.big_img;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Hit h; h.prim = 0; h.obj = objs[i]; h.elem = elems[i]; h.t = 0.0f;
        const V3 nrm = surface_normal(sm, big, v, h, mk(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        out[3 * i] = nrm.x; out[3 * i + 1] = nrm.y; out[3 * i + 2] = nrm.z;
    }
}

// Object::GetRandomPointOnSurface(gen): Light (source/objects/light.h:47-53), SphereObject (source/objects/sphere.h:59-69), base
// Object (source/objects/object.h:26-29); the generator states come in and go back advanced, like the by-reference generator
__global__ __launch_bounds__(kBlock) void k_surface_points(const SceneView v, uint32_t obj, uint64_t n, uint64_t* __restrict__ states, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    stage_scene<false>(v, smem);
    const LdsPtr sm = (LdsPtr)smem;
    const uint32_t orec = v.off_obj + obj * OBJ_STRIDE;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Rng g; g.s0 = states[2 * i]; g.s1 = states[2 * i + 1];
        const V3 p = light_point(sm, orec, g);
        states[2 * i] = g.s0; states[2 * i + 1] = g.s1;
        out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
    }
}

// Material::BRDF(incident, reflection, normal), source/core/material.h:24-53, through the kernels' own brdf2 (the material's
// fields are laid out as the object record brdf2 reads)
__global__ __launch_bounds__(kBlock) void k_brdf(uint32_t diffuse, float albedo, float on_a, float on_b, uint64_t n, const float* __restrict__ inc,
                                                 const float* __restrict__ refl, const float* __restrict__ nrm, double* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint32_t rec[OBJ_STRIDE];
    if (threadIdx.x == 0) { rec[OBJ_DIFFUSE] = diffuse; rec[OBJ_ALBEDO] = __float_as_uint(albedo); rec[OBJ_ON_A] = __float_as_uint(on_a); rec[OBJ_ON_B] = __float_as_uint(on_b); }
    __syncthreads();
    const LdsPtr sm = (LdsPtr)rec;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double f0, f1;
        brdf2(sm, 0u, mk(inc[3 * i], inc[3 * i + 1], inc[3 * i + 2]), mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), true,
              mk(refl[3 * i], refl[3 * i + 1], refl[3 * i + 2]), false, mk(0, 0, 0), f0, f1);
        out[i] = f0;
    }
}

// source/main.cpp:142-164 + GLImageBuffer::SetPixel (source/opengl/data.cpp:28-50).  pow / exp are the fixed-order pm_pow / pm_exp:
// the bytes equal the oracle's (portable mode) on every input, and libm's wherever 255 * pow is not within ~1e-13 of an integer.
__global__ __launch_bounds__(kBlock) void k_resolve(const double* __restrict__ sums, uint64_t npix, double spp, int tonemap, double inv_gamma,
                                                    double exposure, uint8_t* __restrict__ out_bgra, double* __restrict__ out_mean)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        double c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double x = sums[3 * p + k] / spp;
            if (out_mean) out_mean[3 * p + k] = x;
            if (tonemap == MCRT_TONEMAP_REINHARD) {
                x = x / (x + 1.0);
                x = pm_pow(x, inv_gamma);
            } else if (tonemap == MCRT_TONEMAP_EXPOSURE) {
                x = 1.0 - pm_exp(-x * exposure);
                x = pm_pow(x, inv_gamma);
            }
            double t = (x < 1.0) ? x : 1.0;
            t = (t < 0.0) ? 0.0 : t;
            c[k] = t;
        }
        if (out_bgra) {
            uchar4 px;
            px.x = (unsigned char)(c[2] * 255.0);
            px.y = (unsigned char)(c[1] * 255.0);
            px.z = (unsigned char)(c[0] * 255.0);
            px.w = (unsigned char)(1.0 * 255.0);
            reinterpret_cast<uchar4*>(out_bgra)[p] = px;
        }
    }
}

thread_local std::string g_create_error;

// Staging area of mcrt_intersect_noalloc: device memory the library never allocates at run time.  One copy per GPU
// (the code object is loaded once per device); every context on a device shares that device's copy.  In host code the
// NAME is only a handle for hipGetSymbolAddress - the kernels get the device address looked up in mcrt_create.
constexpr size_t kNoAllocSlot = ((size_t)MCRT_NOALLOC_MAX_RAYS * 12u + 255u) & ~(size_t)255u;   // one 3-float array, 256-byte aligned
constexpr size_t kNoAllocSlotSmall = ((size_t)MCRT_NOALLOC_MAX_RAYS * 4u + 255u) & ~(size_t)255u;
constexpr size_t kNoAllocBytes = 2 * kNoAllocSlot + 3 * kNoAllocSlotSmall;                      // origins, dirs | obj, elem, t
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

} // namespace

// ===================================================================== context
struct mcrt_ctx {
    int device = 0;
    int cu_count = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    FlatScene flat;
    SceneView view{};
    uint32_t* d_small = nullptr;
    uint32_t* d_big = nullptr;
    bool have_scene = false;
    bool tree_on_host = true;                   // flat.node_box / node_words / refs mirror the device tree (false after a device build, until downloaded)
    float last_build_ms = 0.0f;                 // device build: kernel time of the last one
    bool big_in_lds = false;
    size_t lds_bytes = 0;

    CameraDev cam{};
    bool have_camera = false;
    double* d_accum_own = nullptr;
    size_t accum_pixels = 0;
    double* d_accum = nullptr;

    int accel_request = MCRT_ACCEL_AUTO;   // what the caller asked for
    int accel = MCRT_ACCEL_OCTREE;         // what kernels run with (AUTO resolved per scene)
    bool counting = false;
    unsigned long long* d_counters = nullptr;   // 8 x u64
    uint32_t* d_tile = nullptr;                 // 2 x u32
    unsigned long long* d_dropped = nullptr;    // dropped (non-finite) samples since the last clear
    uint32_t grid_mult = 4;                     // persistent workgroups per CU (MCRT_GRID_MULT, tuning knob)
    uint32_t* d_pend = nullptr;
    void* d_park = nullptr;                     // parked estimator state of the render kernel's lanes (octree walked from HBM)
    size_t park_bytes = 0;
    size_t pend_words = 0;
    uint32_t grid_blocks = 0;
    uint32_t lanes_override = 0;                // mcrt_set_lanes_per_pixel (0: automatic)
    uint8_t* d_resolve_img = nullptr;           // mcrt_resolve staging, grown on demand
    double* d_resolve_mean = nullptr;
    size_t resolve_img_pixels = 0, resolve_mean_pixels = 0;
    double* d_partial = nullptr;                // slice sums of the current pass, [slices][pixels][3]
    size_t partial_bytes = 0;
    int min_trace_override = -1;                // MCRT_MIN_TRACE (tuning knob): LanePolicy::min_trace
    int fetch_min_override = -1;                // MCRT_FETCH_MIN (tuning knob): LanePolicy::fetch_min
    int refill_override = -1;                   // MCRT_REFILL (tuning knob): LanePolicy::refill, threshold
    uint32_t wave_slices = 64;                  // MCRT_WAVE_SLICES (tuning knob): slices of a pixel that share a wavefront, at most

    // wavefront pipeline (k_wf_shade / k_wf_trace): path-slot state, hits, the two slot queues, counters
    int wavefront = -1;                         // MCRT_WAVEFRONT (tuning knob): 1 / 0 forces it on / off for HBM-resident octree scenes, -1: automatic
    uint32_t wf_slots_override = 0;             // MCRT_WF_SLOTS (tuning knob)
    int wf_trace_variant = 0;                   // MCRT_WF_TRACE (tuning knob): 0: 6 blocks per CU, 1: 8 blocks per CU, 2: 4 blocks per CU
    U4* d_wf_state = nullptr;
    U4* d_wf_hits = nullptr;
    uint32_t* d_wf_queue = nullptr;             // 2 x slots
    uint32_t* d_wf_cnt = nullptr;
    uint32_t* h_wf_cnt = nullptr;               // pinned: the queue length read back every few iterations
    uint32_t wf_slots = 0, wf_chunks = 0;
    uint32_t* d_wf_pend = nullptr;
    size_t wf_pend_words = 0;
    uint32_t wf_iterations = 0;                 // of the last render (introspection)

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    unsigned char* noalloc_scratch = nullptr;   // this device's copy of `scratch` (hipGetSymbolAddress in mcrt_create)
};

namespace {

int fail(mcrt_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(ctx, MCRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int bind_device(mcrt_ctx* c) { HIP_TRY(c, hipSetDevice(c->device)); return MCRT_OK; }

size_t flat_lds_bytes(const mcrt_ctx* c);
size_t lds_for(const mcrt_ctx* c)
{
    if (c->accel == MCRT_ACCEL_SCAN) return 0;
    if (c->accel == MCRT_ACCEL_FLAT) return flat_lds_bytes(c);
    return c->lds_bytes + ((c->accel == MCRT_ACCEL_OCTREE && !c->big_in_lds) ? kWalkLdsWords * 4u : 0u);
}

// the flat scan keeps the triangle records and its per-wavefront work areas in LDS and packs (object, element) into 32 bits
size_t flat_lds_bytes(const mcrt_ctx* c) { return (size_t)c->view.ntri * TRI_STRIDE * 4u + 16u + (kBlock / 64) * FLAT_AREA_WORDS * 4u; }
bool flat_ok(const mcrt_ctx* c)
{
    return c->view.has_pairs && c->view.ntri < (1u << FLAT_ELEM_BITS) && c->view.nobj < 4095u && flat_lds_bytes(c) <= 40u * 1024u;   // 4 blocks per CU
}
// what the kernels of the current accel mode get: the flat scan addresses triangles in its LDS copy
SceneView view_for(const mcrt_ctx* c)
{
    SceneView v = c->view;
    if (c->accel == MCRT_ACCEL_FLAT) v.off_tri = 0;
    return v;
}

void resolve_accel(mcrt_ctx* c)
{
    c->accel = c->accel_request;
    if (c->accel == MCRT_ACCEL_AUTO)
        c->accel = (c->have_scene && c->view.nprim <= MCRT_SCAN_MAX_PRIMS) ? (flat_ok(c) ? MCRT_ACCEL_FLAT : MCRT_ACCEL_SCAN) : MCRT_ACCEL_OCTREE;
}

// persistent grid: the 256-lane workgroups that are resident together - every tracing kernel allocates 128 VGPRs, which is
// 4 wavefronts per SIMD, 4 workgroups per CU; more would only queue behind them (and double the parked-branch stack)
uint32_t grid_for(const mcrt_ctx* c) { return (uint32_t)c->cu_count * c->grid_mult; }

bool scene_has_refractive(const mcrt_ctx* c);
// The parked-branch stacks: one 64-byte line per (launched lane, entry), max_depth entries per lane - a refractive split parks one
// branch and both branches are one level deeper, so max_depth bounds the stack (source/scene.cpp:209-214).  1.7 GB at the reference's
// depth of 100 with 262144 launched lanes; a scene without refractive surfaces never parks anything and gets no stacks.
int ensure_pend(mcrt_ctx* c, uint32_t max_depth)
{
    const uint32_t entries = scene_has_refractive(c) ? std::max(1u, max_depth) : 1u;
    const size_t lanes = (size_t)grid_for(c) * kBlock;
    const size_t words = lanes * kPendFields * entries;
    if (words > c->pend_words) {
        if (c->d_pend) { HIP_TRY(c, hipFree(c->d_pend)); c->d_pend = nullptr; c->pend_words = 0; }
        HIP_TRY(c, hipMalloc(&c->d_pend, words * sizeof(uint32_t)));
        c->pend_words = words;
    }
    return MCRT_OK;
}

// The parked estimator state (trace_lane_parked, ParkIO): 20 chunks of 16 bytes per launched lane.
int ensure_park(mcrt_ctx* c)
{
    const size_t bytes = (size_t)grid_for(c) * kBlock * kParkChunksMulti * 16u;
    if (bytes > c->park_bytes) {
        if (c->d_park) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->d_park)); c->d_park = nullptr; c->park_bytes = 0; }
        HIP_TRY(c, hipMalloc(&c->d_park, bytes));
        c->park_bytes = bytes;
    }
    return MCRT_OK;
}

template <class K> int set_lds_limit(mcrt_ctx* c, K kernel, size_t bytes)
{
    if (bytes > 48 * 1024) HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MCRT_OK;
}

// One-light kernels (every BASELINE scene) exist for the timed builds; counting builds and scenes with several lights use the general ones.
template <bool kBigLds, int kAccel, bool kCount> int launch_render_t(mcrt_ctx* c, const RenderArgs& a)
{
    auto kern = (kCount || c->view.nlights > 1) ? k_render<kBigLds, kAccel, kCount, true> : k_render<kBigLds, kAccel, kCount, kCount>;
    const size_t lds = lds_for(c);
    int rc = set_lds_limit(c, kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(c)), dim3(kBlock), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}
template <bool kBigLds, int kAccel, bool kCount> int launch_list_t(mcrt_ctx* c, const ListArgs& a)
{
    auto kern = (kCount || c->view.nlights > 1) ? k_trace_list<kBigLds, kAccel, kCount, true> : k_trace_list<kBigLds, kAccel, kCount, kCount>;
    int rc = set_lds_limit(c, kern, lds_for(c));
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(c)), dim3(kBlock), lds_for(c), c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}
template <bool kBigLds, int kAccel, bool kCount>
int launch_intersect_t(mcrt_ctx* c, uint64_t n, const float* o, const float* d, int32_t* obj, uint32_t* elem, float* t)
{
    auto kern = k_intersect<kBigLds, kAccel, kCount>;
    int rc = set_lds_limit(c, kern, lds_for(c));
    if (rc) return rc;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)grid_for(c));
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kBlock), lds_for(c), c->stream, view_for(c), n, o, d, obj, elem, t, c->d_counters);
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}
template <bool kBigLds, int kAccel> int launch_unlit_t(mcrt_ctx* c, double* out)
{
    auto kern = k_unlit<kBigLds, kAccel>;
    int rc = set_lds_limit(c, kern, lds_for(c));
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(c)), dim3(kBlock), lds_for(c), c->stream, view_for(c), c->cam, out);
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}

// Dispatch on (scene residency, accel mode, counting).  Counting builds exist for the octree only.
#define MCRT_DISPATCH(c, CALL_, ...)                                                                                   \
    do {                                                                                                                \
        const bool lds_ = (c)->big_in_lds;                                                                              \
        if ((c)->accel == MCRT_ACCEL_SCAN) {                                                                            \
            if ((c)->counting) return CALL_<false, MCRT_ACCEL_SCAN, true>(__VA_ARGS__);                                \
            return CALL_<false, MCRT_ACCEL_SCAN, false>(__VA_ARGS__);                                                   \
        }                                                                                                               \
        if ((c)->accel == MCRT_ACCEL_FLAT) {                                                                            \
            if ((c)->counting) return CALL_<true, MCRT_ACCEL_FLAT, true>(__VA_ARGS__);                                 \
            return CALL_<true, MCRT_ACCEL_FLAT, false>(__VA_ARGS__);                                                    \
        }                                                                                                               \
        if ((c)->accel == MCRT_ACCEL_BRUTE) {                                                                           \
            if ((c)->counting) return lds_ ? CALL_<true, MCRT_ACCEL_BRUTE, true>(__VA_ARGS__) : CALL_<false, MCRT_ACCEL_BRUTE, true>(__VA_ARGS__);   \
            return lds_ ? CALL_<true, MCRT_ACCEL_BRUTE, false>(__VA_ARGS__) : CALL_<false, MCRT_ACCEL_BRUTE, false>(__VA_ARGS__);                    \
        }                                                                                                               \
        if ((c)->counting) return lds_ ? CALL_<true, MCRT_ACCEL_OCTREE, true>(__VA_ARGS__) : CALL_<false, MCRT_ACCEL_OCTREE, true>(__VA_ARGS__);     \
        return lds_ ? CALL_<true, MCRT_ACCEL_OCTREE, false>(__VA_ARGS__) : CALL_<false, MCRT_ACCEL_OCTREE, false>(__VA_ARGS__);                      \
    } while (0)

int launch_render(mcrt_ctx* c, const RenderArgs& a) { MCRT_DISPATCH(c, launch_render_t, c, a); }
int launch_list(mcrt_ctx* c, const ListArgs& a) { MCRT_DISPATCH(c, launch_list_t, c, a); }
int launch_intersect(mcrt_ctx* c, uint64_t n, const float* o, const float* d, int32_t* obj, uint32_t* elem, float* t)
{
    MCRT_DISPATCH(c, launch_intersect_t, c, n, o, d, obj, elem, t);
}
int launch_unlit(mcrt_ctx* c, double* out)
{
    if (c->accel == MCRT_ACCEL_SCAN) return launch_unlit_t<false, MCRT_ACCEL_SCAN>(c, out);
    if (c->accel == MCRT_ACCEL_FLAT) return launch_unlit_t<true, MCRT_ACCEL_FLAT>(c, out);
    if (c->accel == MCRT_ACCEL_BRUTE) return c->big_in_lds ? launch_unlit_t<true, MCRT_ACCEL_BRUTE>(c, out) : launch_unlit_t<false, MCRT_ACCEL_BRUTE>(c, out);
    return c->big_in_lds ? launch_unlit_t<true, MCRT_ACCEL_OCTREE>(c, out) : launch_unlit_t<false, MCRT_ACCEL_OCTREE>(c, out);
}

// ---- wavefront pipeline, host side
// Used for octree scenes whose image is walked from HBM (the 1M-triangle configuration).  Slots: 2^21 path slots (512 MB of
// state), fewer when the scene has refractive surfaces: every slot then owns a parked-branch stack of max_depth 64-byte
// lines, and the stacks together are held to 8 GB.
bool use_wavefront(const mcrt_ctx* c)
{
    if (c->accel != MCRT_ACCEL_OCTREE || c->big_in_lds) return false;
    // opt-in (MCRT_WAVEFRONT=1): measured 175 ms against k_render's 103 ms on config4_1024_16 - the walk itself runs at 13 of 64
    // lanes per instruction whether or not finished lanes are refilled at once (profiles/r02_wavefront_experiment.txt)
    return c->wavefront == 1;
}
bool scene_has_refractive(const mcrt_ctx* c)
{
    for (uint32_t i = 0; i < c->view.nobj; ++i)
        if (c->flat.small_img[c->view.off_obj + (size_t)i * OBJ_STRIDE + OBJ_SURFACE] == (uint32_t)MCRT_SURFACE_REFRACTIVE) return true;
    return false;
}
int ensure_wavefront(mcrt_ctx* c, uint32_t max_depth)
{
    const bool refr = scene_has_refractive(c);
    uint32_t slots = c->wf_slots_override ? c->wf_slots_override : (1u << 21);
    if (refr) {
        const uint64_t per_slot = (uint64_t)std::max(1u, max_depth) * kPendFields * 4u;
        while (slots > 65536u && (uint64_t)slots * per_slot > (8ull << 30)) slots >>= 1;
    }
    slots = (slots + 63u) & ~63u;
    const uint32_t chunks = c->view.nlights > 1 ? kWfChunksMulti : kWfChunks;
    if (slots != c->wf_slots || chunks > c->wf_chunks) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->d_wf_state) { (void)hipFree(c->d_wf_state); c->d_wf_state = nullptr; }
        if (c->d_wf_hits) { (void)hipFree(c->d_wf_hits); c->d_wf_hits = nullptr; }
        if (c->d_wf_queue) { (void)hipFree(c->d_wf_queue); c->d_wf_queue = nullptr; }
        c->wf_slots = 0;
        HIP_TRY(c, hipMalloc(&c->d_wf_state, (size_t)slots * chunks * sizeof(U4)));
        HIP_TRY(c, hipMalloc(&c->d_wf_hits, (size_t)slots * sizeof(U4)));
        HIP_TRY(c, hipMalloc(&c->d_wf_queue, (size_t)slots * 2 * sizeof(uint32_t)));
        c->wf_slots = slots; c->wf_chunks = chunks;
    }
    if (!c->d_wf_cnt) HIP_TRY(c, hipMalloc(&c->d_wf_cnt, WF_CNT_WORDS * sizeof(uint32_t)));
    if (!c->h_wf_cnt) HIP_TRY(c, hipHostMalloc((void**)&c->h_wf_cnt, WF_CNT_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    const size_t pend_words = refr ? (size_t)slots * std::max(1u, max_depth) * kPendFields : 0;
    if (pend_words > c->wf_pend_words) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->d_wf_pend) { (void)hipFree(c->d_wf_pend); c->d_wf_pend = nullptr; c->wf_pend_words = 0; }
        HIP_TRY(c, hipMalloc(&c->d_wf_pend, pend_words * sizeof(uint32_t)));
        c->wf_pend_words = pend_words;
    }
    return MCRT_OK;
}

// `bound`: an upper bound of the queue length the kernel will find (the queue never grows: a slot that has run out of work does not
// come back), so that the drained tail of a render does not launch - and stage the scene into - thousands of idle workgroups
uint32_t wf_grid(uint32_t full, uint32_t bound) { return std::max(1u, std::min(full, (bound + kBlock - 1u) / kBlock)); }

template <bool kCount> int launch_wf_shade(mcrt_ctx* c, const WfArgs& a, uint32_t bound)
{
    const size_t lds = c->lds_bytes;
    const uint32_t grid = wf_grid(grid_for(c), bound);
    if (c->view.nlights > 1) hipLaunchKernelGGL((k_wf_shade<kCount, true>), dim3(grid), dim3(kBlock), lds, c->stream, a);
    else hipLaunchKernelGGL((k_wf_shade<kCount, false>), dim3(grid), dim3(kBlock), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}
template <bool kCount> int launch_wf_trace(mcrt_ctx* c, const WfArgs& a, uint32_t bound)
{
    // LDS: the SMALL image + the top of every lane's stack; the number of resident blocks follows from the stack entries kept in LDS
    const size_t small_bytes = c->lds_bytes;
    if (c->wf_trace_variant == 1) {
        hipLaunchKernelGGL((k_wf_trace<kCount, 8, 8>), dim3(wf_grid((uint32_t)c->cu_count * 8u, bound)), dim3(kBlock), small_bytes + 8u * 2u * kBlock * 4u, c->stream, a);
    } else if (c->wf_trace_variant == 2) {
        hipLaunchKernelGGL((k_wf_trace<kCount, 18, 4>), dim3(wf_grid((uint32_t)c->cu_count * 4u, bound)), dim3(kBlock), small_bytes + 18u * 2u * kBlock * 4u, c->stream, a);
    } else {
        hipLaunchKernelGGL((k_wf_trace<kCount, 11, 6>), dim3(wf_grid((uint32_t)c->cu_count * 6u, bound)), dim3(kBlock), small_bytes + 11u * 2u * kBlock * 4u, c->stream, a);
    }
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}

// The render pass of `r` through the wavefront pipeline.  Synchronises the stream every few iterations (the host has to learn
// when the slot queue has run empty); the accumulators are complete when it returns.
int render_wavefront(mcrt_ctx* c, const RenderArgs& r)
{
    WfArgs a;
    a.r = r;
    a.r.pend = c->d_wf_pend;
    a.state = c->d_wf_state; a.hits = c->d_wf_hits; a.cnt = c->d_wf_cnt; a.slots = c->wf_slots;
    uint32_t* q[2] = { c->d_wf_queue, c->d_wf_queue + c->wf_slots };
    HIP_TRY(c, hipMemsetAsync(c->d_wf_cnt, 0, WF_CNT_WORDS * sizeof(uint32_t), c->stream));
    // every slot takes its first item and first camera ray
    a.first = 1; a.q_in = nullptr; a.in_idx = WF_CNT_Q1; a.q_out = q[0]; a.out_idx = WF_CNT_Q0;
    uint32_t bound = c->wf_slots;
    int rc = c->counting ? launch_wf_shade<true>(c, a, bound) : launch_wf_shade<false>(c, a, bound);
    if (rc) return rc;
    a.first = 0;
    uint32_t it = 0;
    for (;; ++it) {
        const uint32_t in = it & 1u, out = in ^ 1u;
        a.q_in = q[in]; a.in_idx = in; a.q_out = q[out]; a.out_idx = out;
        rc = c->counting ? launch_wf_trace<true>(c, a, bound) : launch_wf_trace<false>(c, a, bound);
        if (rc) return rc;
        rc = c->counting ? launch_wf_shade<true>(c, a, bound) : launch_wf_shade<false>(c, a, bound);
        if (rc) return rc;
        if ((it & 3u) == 3u) {
            HIP_TRY(c, hipMemcpyAsync(c->h_wf_cnt, c->d_wf_cnt, WF_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->h_wf_cnt[out] == 0u) break;
            bound = c->h_wf_cnt[out];
            if (it > 4000000u) return fail(c, MCRT_ERR_STATE, "wavefront pipeline did not drain");
        }
    }
    c->wf_iterations = it + 1u;
    return MCRT_OK;
}

// Scheduling of the render kernel's lanes (LanePolicy), measured per kernel family:
//   scan / flat / LDS-resident scenes: trace lanes batched from 36 up (profiles/r02_sweep_min_trace.txt), fetches in rounds of 64
//   (smaller fetch groups mix tiles in a wavefront and lose 8 % and more, profiles/r02_sweep_fetch_min.txt);
//   octree over an HBM-resident image: lanes run free (fetch at once, no batching): walks are long and incoherent anyway.
LanePolicy lane_policy(const mcrt_ctx* c)
{
    const bool walk_hbm = c->accel == MCRT_ACCEL_OCTREE && !c->big_in_lds;
    LanePolicy p;
    p.min_trace = c->min_trace_override >= 0 ? (uint32_t)c->min_trace_override : (walk_hbm ? 0u : 36u);
    p.fetch_min = c->fetch_min_override >= 1 ? (uint32_t)c->fetch_min_override : (walk_hbm ? 1u : 64u);
    // measured (profiles/r01_refill_sweep.txt): refilling pays when the image is walked from HBM/L2 (long, exponentially distributed
    // walks); with an LDS-resident image the extra shading passes cost more than they save
    const uint32_t refill = c->refill_override >= 0 ? (uint32_t)c->refill_override : (c->big_in_lds ? 0u : 30u);
    p.refill = refill;
    return p;
}

int need_scene(mcrt_ctx* c, bool camera)
{
    if (!c) return MCRT_ERR_INVALID;
    if (!c->have_scene) return fail(c, MCRT_ERR_STATE, "no scene uploaded (mcrt_upload_scene)");
    if (camera && !c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    return bind_device(c);
}

// RCCL is bound at run time (dlopen), so that the library loads - and every single-GPU entry point works - where librccl is
// absent, and a process that already carries an RCCL (PyTorch ships its own) is not given a second one: symbols already in
// the process are preferred.  MCRT_RCCL_LIB names the library to load instead of librccl.so.1 / librccl.so.  The two enum values
// mcrt_reduce passes are those of rccl.h for every NCCL 2.x API (ncclDouble = 8, ncclSum = 0): the major version is checked once
// through ncclGetVersion, and a library that reports anything else is refused rather than trusted.
struct Rccl {
    typedef int (*InitAll)(void** comms, int ndev, const int* devlist);
    typedef int (*Destroy)(void* comm);
    typedef int (*Reduce)(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, hipStream_t stream);
    typedef const char* (*ErrStr)(int);
    typedef int (*GetVersion)(int*);
    InitAll init_all = nullptr;
    Destroy destroy = nullptr;
    Reduce reduce = nullptr;
    ErrStr err = nullptr;
    int version = 0;
    std::string why;
    bool ok = false;
};
constexpr int kNcclDouble = 8, kNcclSum = 0;       // rccl.h: ncclDataType_t::ncclDouble (= ncclFloat64), ncclRedOp_t::ncclSum
const Rccl& rccl()
{
    static const Rccl r = [] {
        Rccl x;
        void* h = RTLD_DEFAULT;
        if (!dlsym(RTLD_DEFAULT, "ncclReduce")) {
            const char* named = getenv("MCRT_RCCL_LIB");
            if (named && *named) h = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
            else {
                h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            }
            if (!h) {
                const char* m = dlerror();               // once: the call clears the message
                x.why = std::string("librccl not loadable: ") + (m ? m : "?");
                return x;
            }
        }
        x.init_all = (Rccl::InitAll)dlsym(h, "ncclCommInitAll");
        x.destroy = (Rccl::Destroy)dlsym(h, "ncclCommDestroy");
        x.reduce = (Rccl::Reduce)dlsym(h, "ncclReduce");
        x.err = (Rccl::ErrStr)dlsym(h, "ncclGetErrorString");
        if (!(x.init_all && x.destroy && x.reduce)) { x.why = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclReduce"; return x; }
        // NCCL_VERSION_CODE: major * 1000 + ... up to 2.8, major * 10000 + ... from 2.9 on
        Rccl::GetVersion ver = (Rccl::GetVersion)dlsym(h, "ncclGetVersion");
        if (!ver || ver(&x.version) != 0) { x.why = "librccl: ncclGetVersion missing or failing, enum values cannot be checked"; return x; }
        const int major = x.version >= 20000 ? x.version / 10000 : x.version / 1000;
        if (major != 2) { x.why = "librccl reports NCCL API version " + std::to_string(x.version) + ": ncclDouble / ncclSum are known for 2.x only"; return x; }
        x.ok = true;
        return x;
    }();
    return r;
}

template <class T> struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

} // namespace

// ======================================================================== ABI
extern "C" {

#ifdef MCRT_SECTION_PROFILE
// diagnostic builds only: read and clear the section profile (tools/section_profile.py)
int mcrt_debug_section_profile(uint64_t out[32])
{
    static const unsigned long long zero[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(zero)) != hipSuccess) return MCRT_ERR_HIP;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof(zero)) == hipSuccess ? MCRT_OK : MCRT_ERR_HIP;
}
#endif

int mcrt_create(int device_id, mcrt_ctx** out_ctx)
{
    if (!out_ctx) return fail(nullptr, MCRT_ERR_INVALID, "out_ctx is null");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(nullptr, MCRT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= count) return fail(nullptr, MCRT_ERR_NO_DEVICE, "device_id out of range");
    mcrt_ctx* c = new (std::nothrow) mcrt_ctx;
    if (!c) return fail(nullptr, MCRT_ERR_INVALID, "out of host memory");
    c->device = device_id;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        mcrt_destroy(c);
        return MCRT_ERR_HIP;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("MCRT_MIN_TRACE")) { int r = atoi(e); if (r >= 0 && r <= 64) c->min_trace_override = r; }
    if (const char* e = getenv("MCRT_GRID_MULT")) { int r = atoi(e); if (r >= 1 && r <= 16) c->grid_mult = (uint32_t)r; }
    if (const char* e = getenv("MCRT_REFILL")) { int r = atoi(e); if (r >= 0 && r <= 64) c->refill_override = r; }
    if (const char* e = getenv("MCRT_FETCH_MIN")) { int r = atoi(e); if (r >= 1 && r <= 64) c->fetch_min_override = r; }
    if (const char* e = getenv("MCRT_WAVE_SLICES")) { int r = atoi(e); if (r >= 1 && r <= 64) c->wave_slices = (uint32_t)r; }
    if (const char* e = getenv("MCRT_WAVEFRONT")) { int r = atoi(e); if (r == 0 || r == 1) c->wavefront = r; }
    if (const char* e = getenv("MCRT_WF_SLOTS")) { long r = atol(e); if (r >= 4096 && r <= (1l << 24)) c->wf_slots_override = (uint32_t)r; }
    if (const char* e = getenv("MCRT_WF_TRACE")) { int r = atoi(e); if (r >= 0 && r <= 2) c->wf_trace_variant = r; }
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    c->stream = c->own_stream;
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc(&c->d_counters, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->d_tile, 2 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->d_dropped, sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->d_dropped, 0, sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMemset(c->d_counters, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
    // with this context's device current: the address of THIS device's copy of the static staging array
    if ((e = hipGetSymbolAddress((void**)&c->noalloc_scratch, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    *out_ctx = c;
    return MCRT_OK;
}

int mcrt_destroy(mcrt_ctx* c)
{
    if (!c) return MCRT_OK;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->d_big) (void)hipFree(c->d_big);
    if (c->d_accum_own) (void)hipFree(c->d_accum_own);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_tile) (void)hipFree(c->d_tile);
    if (c->d_dropped) (void)hipFree(c->d_dropped);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_resolve_img) (void)hipFree(c->d_resolve_img);
    if (c->d_resolve_mean) (void)hipFree(c->d_resolve_mean);
    if (c->d_pend) (void)hipFree(c->d_pend);
    if (c->d_park) (void)hipFree(c->d_park);
    if (c->d_wf_state) (void)hipFree(c->d_wf_state);
    if (c->d_wf_hits) (void)hipFree(c->d_wf_hits);
    if (c->d_wf_queue) (void)hipFree(c->d_wf_queue);
    if (c->d_wf_cnt) (void)hipFree(c->d_wf_cnt);
    if (c->d_wf_pend) (void)hipFree(c->d_wf_pend);
    if (c->h_wf_cnt) (void)hipHostFree(c->h_wf_cnt);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return MCRT_OK;
}

const char* mcrt_last_error(const mcrt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int mcrt_set_stream(mcrt_ctx* c, void* hip_stream)
{
    if (!c) return MCRT_ERR_INVALID;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return MCRT_OK;
}

namespace {
int upload_scene_impl(mcrt_ctx* c, const mcrt_scene_desc* scene, const mcrt_accel_params* accel, bool device_build)
{
    if (!c) return MCRT_ERR_INVALID;
    int rc = bind_device(c);
    if (rc) return rc;
    FlatScene flat;
    std::string err;
    rc = flatten_scene(scene, accel, flat, err, !device_build);
    if (rc) return fail(c, rc, err);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_small) { HIP_TRY(c, hipFree(c->d_small)); c->d_small = nullptr; }
    if (c->d_big) { HIP_TRY(c, hipFree(c->d_big)); c->d_big = nullptr; }
    c->have_scene = false;
    HIP_TRY(c, hipMalloc(&c->d_small, std::max<size_t>(flat.small_img.size(), 1) * 4));
    HIP_TRY(c, hipMemcpy(c->d_small, flat.small_img.data(), flat.small_img.size() * 4, hipMemcpyHostToDevice));
    if (!device_build) {
        c->last_build_ms = 0.0f;                 // mcrt_last_build_ms reports the device build of the scene IN USE only
        HIP_TRY(c, hipMalloc(&c->d_big, std::max<size_t>(flat.big_img.size(), 1) * 4));
        HIP_TRY(c, hipMemcpy(c->d_big, flat.big_img.data(), flat.big_img.size() * 4, hipMemcpyHostToDevice));
        c->tree_on_host = true;
    } else {
        // the octree is built by kernels (csrc/mcrt_build.h) from the caller's triangles and the SMALL image already on the device;
        // the host flattener has laid out the triangle and pair records from word 0 of its big image
        bld::Params P = { flat.view.nobj, flat.view.nprim, flat.view.ntri, flat.max_leaf, flat.max_depth, flat.stay, flat.pad_rel };
        bld::Result R;
        c->last_build_ms = 0.0f;
        if (bld::build_on_device(c->stream, P, c->d_small, flat.view.off_obj, scene->triangle_count ? &scene->triangles[0].v0[0] : nullptr, flat.big_img,
                                 flat.view.off_pair, R, err)) {
            if (R.d_big) (void)hipFree(R.d_big);
            return fail(c, MCRT_ERR_HIP, err);
        }
        c->d_big = R.d_big;
        SceneView& v = flat.view;
        v.nnode = R.nnode; v.off_node = R.off_node; v.off_leaf = R.off_leaf; v.off_tri = R.off_tri; v.tri_src = R.off_tri; v.off_pair = R.off_pair;
        v.big_words = (uint32_t)R.big_words;
        for (int a = 0; a < 3; ++a) v.root_size[a] = R.root_cell[3 + a] - R.root_cell[a];
        v.walk_pad = kWalkPadFraction * R.pad;
        v.far_limit = R.pad * kFarOriginFactor;
        flat.tree_depth = R.depth; flat.ref_count = R.nrefs; flat.pad = R.pad;
        for (int a = 0; a < 6; ++a) flat.root_cell[a] = R.root_cell[a];
        flat.big_img.clear(); flat.big_img.shrink_to_fit();
        c->tree_on_host = false;
        c->last_build_ms = R.build_ms;
    }
    c->flat = std::move(flat);
    c->view = c->flat.view;
    c->view.small_img = c->d_small;
    c->view.big_img = c->d_big;
    const size_t small_bytes = (((size_t)c->view.small_words + 3u) & ~size_t(3)) * 4;
    const size_t all_bytes = small_bytes + (size_t)c->view.big_words * 4;
    if (small_bytes > kLdsBudgetBytes) return fail(c, MCRT_ERR_LIMIT, "object table does not fit the LDS budget");
    c->big_in_lds = all_bytes <= kLdsBudgetBytes;
    c->lds_bytes = c->big_in_lds ? all_bytes : small_bytes;
    c->have_scene = true;
    if (c->accel_request == MCRT_ACCEL_SCAN && !c->view.has_pairs) c->accel_request = MCRT_ACCEL_AUTO;   // scan cannot serve this scene
    if (c->accel_request == MCRT_ACCEL_FLAT && !flat_ok(c)) c->accel_request = MCRT_ACCEL_AUTO;
    resolve_accel(c);
    return MCRT_OK;
}

// after a device build the host copies of the tree (mcrt_octree_download, introspection) are read back from the records on demand
int ensure_tree_on_host(mcrt_ctx* c)
{
    if (c->tree_on_host) return MCRT_OK;
    int rc = bind_device(c);
    if (rc) return rc;
    const uint32_t nn = c->view.nnode, nr = c->flat.ref_count;
    DevBuf<float> box; DevBuf<uint32_t> words, refs;
    HIP_TRY(c, hipMalloc(&box.p, std::max<size_t>(nn, 1) * 24)); HIP_TRY(c, hipMalloc(&words.p, std::max<size_t>(nn, 1) * 8)); HIP_TRY(c, hipMalloc(&refs.p, std::max<size_t>(nr, 1) * 4));
    const uint32_t n = std::max(nn, nr);
    hipLaunchKernelGGL(bld::k_bld_unpack, dim3(bld::blocks_for(n)), dim3(bld::kThreads), 0, c->stream, nn, nr, (const uint32_t*)c->d_big, c->view.off_node, c->view.off_leaf, box.p,
                       words.p, refs.p);
    HIP_TRY(c, hipGetLastError());
    c->flat.node_box.resize((size_t)nn * 6); c->flat.node_words.resize((size_t)nn * 2); c->flat.refs.resize(nr);
    HIP_TRY(c, hipMemcpyAsync(c->flat.node_box.data(), box.p, (size_t)nn * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->flat.node_words.data(), words.p, (size_t)nn * 8, hipMemcpyDeviceToHost, c->stream));
    if (nr) HIP_TRY(c, hipMemcpyAsync(c->flat.refs.data(), refs.p, (size_t)nr * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tree_on_host = true;
    return MCRT_OK;
}
} // namespace

int mcrt_upload_scene(mcrt_ctx* c, const mcrt_scene_desc* scene, const mcrt_accel_params* accel) { return upload_scene_impl(c, scene, accel, false); }

int mcrt_upload_scene_device_build(mcrt_ctx* c, const mcrt_scene_desc* scene, const mcrt_accel_params* accel) { return upload_scene_impl(c, scene, accel, true); }

int mcrt_refit_scene(mcrt_ctx* c, const mcrt_scene_desc* scene, uint32_t moved_count, const uint32_t* moved_objects, int* out_rebuilt)
{
    if (out_rebuilt) *out_rebuilt = 0;
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (moved_count && !moved_objects) return fail(c, MCRT_ERR_INVALID, "null list of moved objects");
    const mcrt_accel_params accel = { c->flat.max_leaf, c->flat.max_depth, c->flat.pad_rel };      // the parameters the tree in use was built with
    auto rebuild = [&]() {
        if (out_rebuilt) *out_rebuilt = 1;
        return upload_scene_impl(c, scene, &accel, true);
    };
    FlatScene flat;
    std::string err;
    rc = flatten_scene(scene, &accel, flat, err, false);
    if (rc) return fail(c, rc, err);
    // same objects, same kinds, same triangle ranges, same light list: otherwise this is another scene
    const SceneView& v = c->view;
    bool same = flat.view.nobj == v.nobj && flat.view.ntri == v.ntri && flat.view.nprim == v.nprim && flat.view.nlights == v.nlights &&
                flat.view.small_words == v.small_words && flat.view.has_pairs == v.has_pairs && flat.view.off_pair == v.off_pair - v.off_tri;
    for (uint32_t i = 0; same && i < v.nobj; ++i) {
        const uint32_t* a = flat.small_img.data() + flat.view.off_obj + (size_t)i * OBJ_STRIDE;
        const uint32_t* b = c->flat.small_img.data() + v.off_obj + (size_t)i * OBJ_STRIDE;
        same = a[OBJ_KIND] == b[OBJ_KIND] && a[OBJ_FIRST_TRI] == b[OBJ_FIRST_TRI] && a[OBJ_TRI_COUNT] == b[OBJ_TRI_COUNT] && a[OBJ_FIRST_PRIM] == b[OBJ_FIRST_PRIM];
        // a sphere that crosses the "never intersects" radius (source/objects/sphere.h:18) enters or leaves the tree
        if (same && a[OBJ_KIND] == (uint32_t)MCRT_OBJECT_SPHERE)
            same = (__builtin_bit_cast(float, a[OBJ_RADIUS]) >= 1.1920928955078125e-07f) == (__builtin_bit_cast(float, b[OBJ_RADIUS]) >= 1.1920928955078125e-07f);
    }
    for (uint32_t k = 0; same && k < v.nlights; ++k) same = flat.small_img[flat.view.off_lights + k] == c->flat.small_img[v.off_lights + k];
    if (!same) return rebuild();
    std::vector<uint8_t> moved(v.nobj, 0);
    for (uint32_t k = 0; k < moved_count; ++k) {
        if (moved_objects[k] >= v.nobj) return fail(c, MCRT_ERR_INVALID, "moved object index outside the scene");
        moved[moved_objects[k]] = 1;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_small, flat.small_img.data(), flat.small_img.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (!flat.big_img.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_big + v.off_tri, flat.big_img.data(), flat.big_img.size() * 4, hipMemcpyHostToDevice, c->stream));
    bld::Params P = { v.nobj, v.nprim, v.ntri, c->flat.max_leaf, c->flat.max_depth, c->flat.stay, c->flat.pad_rel };
    uint32_t why = 0;
    float ms = 0.0f;
    const int r = bld::refit_on_device(c->stream, P, c->d_small, v.off_obj, scene->triangle_count ? &scene->triangles[0].v0[0] : nullptr, moved, c->d_big, v.off_node,
                                       v.off_leaf, v.off_tri, v.nnode, c->flat.ref_count, c->flat.tree_depth, c->flat.root_cell, c->flat.pad, why, ms, err);
    if (r == 1) return fail(c, MCRT_ERR_HIP, err);
    if (r == 2) return rebuild();          // a moved primitive left the cells that reference it (the triangle records were overwritten: the rebuild rewrites everything)
    c->flat.small_img = flat.small_img;
    c->last_build_ms = ms;
    c->tree_on_host = false;               // the host copies of the node boxes are stale: read back on demand
    return MCRT_OK;
}

int mcrt_last_build_ms(mcrt_ctx* c, float* out_ms)
{
    if (!c || !out_ms) return MCRT_ERR_INVALID;
    if (!c->have_scene || c->last_build_ms <= 0.0f) return fail(c, MCRT_ERR_STATE, "no scene built on the device yet");
    *out_ms = c->last_build_ms;
    return MCRT_OK;
}

int mcrt_set_camera(mcrt_ctx* c, const mcrt_camera* cam)
{
    if (!c || !cam) return MCRT_ERR_INVALID;
    if (cam->width == 0 || cam->height == 0 || (uint64_t)cam->width * cam->height > 0x7FFFFFFFull / 3) return fail(c, MCRT_ERR_INVALID, "bad image size");
    int rc = bind_device(c);
    if (rc) return rc;
    CameraDev& d = c->cam;
    for (int i = 0; i < 3; ++i) d.pos[i] = cam->position[i];
    d.fov_scale = cam->fov_scale;
    for (int i = 0; i < 16; ++i) d.view[i] = cam->view[i];
    d.width = cam->width; d.height = cam->height;
    // PixelBuffer ctor, source/core/pixelbuffer.cpp:20-22
    d.dx = 2.0 / (double)cam->width;
    d.dy = 2.0 / (double)cam->height;
    d.aspect_f = (float)(cam->width / (double)cam->height);
    const size_t px = (size_t)cam->width * cam->height;
    if (px != c->accum_pixels) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->d_accum_own) { HIP_TRY(c, hipFree(c->d_accum_own)); c->d_accum_own = nullptr; }
        HIP_TRY(c, hipMalloc(&c->d_accum_own, px * 3 * sizeof(double)));
        c->accum_pixels = px;
        c->d_accum = c->d_accum_own;         // a caller-bound buffer had the old size: back to the context's own (include/mcrt.h)
    }
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, px * 3 * sizeof(double), c->stream));     // the accumulator IN USE, bound or own
    HIP_TRY(c, hipMemsetAsync(c->d_dropped, 0, sizeof(unsigned long long), c->stream));
    c->have_camera = true;
    return MCRT_OK;
}

int mcrt_camera_look_at(uint32_t width, uint32_t height, float fov_y_degrees, const float position[3], const float look_at[3],
                        const float up[3], mcrt_camera* out)
{
    if (!position || !look_at || !up || !out) return fail(nullptr, MCRT_ERR_INVALID, "null argument");
    if (width == 0 || height == 0) return fail(nullptr, MCRT_ERR_INVALID, "bad image size");
    camera_look_at(width, height, fov_y_degrees, position, look_at, up, out);
    return MCRT_OK;
}

int mcrt_set_accel(mcrt_ctx* c, int mode)
{
    if (!c) return MCRT_ERR_INVALID;
    if (mode < MCRT_ACCEL_OCTREE || mode > MCRT_ACCEL_FLAT) return fail(c, MCRT_ERR_INVALID, "unknown accel mode");
    if (mode == MCRT_ACCEL_FLAT && c->have_scene && !flat_ok(c)) return fail(c, MCRT_ERR_LIMIT, "scene too large for the flat scan (LDS-resident scenes only)");
    if (mode == MCRT_ACCEL_SCAN && c->have_scene && !c->view.has_pairs)
        return fail(c, MCRT_ERR_LIMIT, "scene has more than MCRT_SCAN_PAIR_LIMIT triangles: no scan records, use the octree");
    c->accel_request = mode;
    resolve_accel(c);
    return MCRT_OK;
}

int mcrt_camera_rays(mcrt_ctx* c, uint64_t n, const float* xs, const float* ys, float* out_o, float* out_d)
{
    if (!c) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    if (n == 0) return MCRT_OK;
    if (!xs || !ys || !out_o || !out_d) return fail(c, MCRT_ERR_INVALID, "null buffer");
    int rc = bind_device(c);
    if (rc) return rc;
    DevBuf<float> dx, dy, dor, ddi;
    HIP_TRY(c, hipMalloc(&dx.p, n * 4)); HIP_TRY(c, hipMalloc(&dy.p, n * 4));
    HIP_TRY(c, hipMalloc(&dor.p, n * 12)); HIP_TRY(c, hipMalloc(&ddi.p, n * 12));
    HIP_TRY(c, hipMemcpyAsync(dx.p, xs, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dy.p, ys, n * 4, hipMemcpyHostToDevice, c->stream));
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + kBlock - 1) / kBlock, 65535);
    hipLaunchKernelGGL(k_camera_rays, dim3(blocks), dim3(kBlock), 0, c->stream, c->cam, n, dx.p, dy.p, dor.p, ddi.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_o, dor.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_d, ddi.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_intersect_device(mcrt_ctx* c, uint64_t n, const float* d_o, const float* d_d, int32_t* d_obj, uint32_t* d_elem, float* d_t)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (n == 0) return MCRT_OK;
    if (!d_o || !d_d || !d_obj || !d_elem || !d_t) return fail(c, MCRT_ERR_INVALID, "null buffer");
    if (c->counting) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
    return launch_intersect(c, n, d_o, d_d, d_obj, d_elem, d_t);
}

int mcrt_intersect(mcrt_ctx* c, uint64_t n, const float* o, const float* d, int32_t* out_obj, uint32_t* out_elem, float* out_t)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (n == 0) return MCRT_OK;
    if (!o || !d || !out_obj || !out_elem || !out_t) return fail(c, MCRT_ERR_INVALID, "null buffer");
    DevBuf<float> dor, ddi, dt;
    DevBuf<int32_t> dobj;
    DevBuf<uint32_t> delem;
    HIP_TRY(c, hipMalloc(&dor.p, n * 12)); HIP_TRY(c, hipMalloc(&ddi.p, n * 12)); HIP_TRY(c, hipMalloc(&dt.p, n * 4));
    HIP_TRY(c, hipMalloc(&dobj.p, n * 4)); HIP_TRY(c, hipMalloc(&delem.p, n * 4));
    HIP_TRY(c, hipMemcpyAsync(dor.p, o, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ddi.p, d, n * 12, hipMemcpyHostToDevice, c->stream));
    rc = mcrt_intersect_device(c, n, dor.p, ddi.p, dobj.p, delem.p, dt.p);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_obj, dobj.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_elem, delem.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_t, dt.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_intersect_noalloc(mcrt_ctx* c, uint64_t n, const float* o, const float* d, int32_t* out_obj, uint32_t* out_elem, float* out_t)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (n == 0) return MCRT_OK;
    if (!o || !d || !out_obj || !out_elem || !out_t) return fail(c, MCRT_ERR_INVALID, "null buffer");
    if (n > MCRT_NOALLOC_MAX_RAYS)
        return fail(c, MCRT_ERR_LIMIT, "mcrt_intersect_noalloc: more than MCRT_NOALLOC_MAX_RAYS (" + std::to_string(MCRT_NOALLOC_MAX_RAYS) +
                                           ") rays in one call; split the batch or use mcrt_intersect");
    if (!c->noalloc_scratch) return fail(c, MCRT_ERR_STATE, "static staging area not resolved");
    unsigned char* base = c->noalloc_scratch;      // never the array's name: that is a host-side stand-in
    float* dor = reinterpret_cast<float*>(base);
    float* ddi = reinterpret_cast<float*>(base + kNoAllocSlot);
    int32_t* dobj = reinterpret_cast<int32_t*>(base + 2 * kNoAllocSlot);
    uint32_t* delem = reinterpret_cast<uint32_t*>(base + 2 * kNoAllocSlot + kNoAllocSlotSmall);
    float* dt = reinterpret_cast<float*>(base + 2 * kNoAllocSlot + 2 * kNoAllocSlotSmall);
    // nothing in the staging area has to start at zero: the kernel writes all three outputs of every ray i < n,
    // and only those are copied back
    HIP_TRY(c, hipMemcpyAsync(dor, o, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ddi, d, n * 12, hipMemcpyHostToDevice, c->stream));
    rc = mcrt_intersect_device(c, n, dor, ddi, dobj, delem, dt);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_obj, dobj, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_elem, delem, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_t, dt, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

namespace {
uint32_t batch_blocks(const mcrt_ctx* c, uint64_t n) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)grid_for(c))); }
size_t small_lds_bytes(const mcrt_ctx* c) { return (((size_t)c->view.small_words + 3u) & ~size_t(3)) * 4; }
}

int mcrt_object_intersect(mcrt_ctx* c, uint32_t object_index, uint64_t n, const float* o, const float* d, uint8_t* out_hit, uint32_t* out_elem, float* out_t)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (object_index >= c->view.nobj) return fail(c, MCRT_ERR_INVALID, "object index outside the scene");
    if (n == 0) return MCRT_OK;
    if (!o || !d || !out_hit || !out_elem || !out_t) return fail(c, MCRT_ERR_INVALID, "null buffer");
    DevBuf<float> dor, ddi, dt;
    DevBuf<uint8_t> dhit;
    DevBuf<uint32_t> delem;
    HIP_TRY(c, hipMalloc(&dor.p, n * 12)); HIP_TRY(c, hipMalloc(&ddi.p, n * 12)); HIP_TRY(c, hipMalloc(&dt.p, n * 4));
    HIP_TRY(c, hipMalloc(&dhit.p, n)); HIP_TRY(c, hipMalloc(&delem.p, n * 4));
    HIP_TRY(c, hipMemcpyAsync(dor.p, o, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ddi.p, d, n * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_object_intersect, dim3(batch_blocks(c, n)), dim3(kBlock), small_lds_bytes(c), c->stream, c->view, object_index, n, dor.p, ddi.p, dhit.p, delem.p, dt.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_hit, dhit.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_elem, delem.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_t, dt.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_surface_normals(mcrt_ctx* c, uint64_t n, const uint32_t* objects, const uint32_t* elems, const float* positions, float* out_normals)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (n == 0) return MCRT_OK;
    if (!objects || !elems || !positions || !out_normals) return fail(c, MCRT_ERR_INVALID, "null buffer");
    for (uint64_t i = 0; i < n; ++i) {
        if (objects[i] >= c->view.nobj) return fail(c, MCRT_ERR_INVALID, "object index outside the scene");
        const uint32_t* rec = c->flat.small_img.data() + c->view.off_obj + (size_t)objects[i] * OBJ_STRIDE;
        if (rec[OBJ_KIND] != (uint32_t)MCRT_OBJECT_SPHERE && elems[i] >= rec[OBJ_TRI_COUNT]) return fail(c, MCRT_ERR_INVALID, "element index outside the mesh");
    }
    DevBuf<uint32_t> dob, del;
    DevBuf<float> dpos, dout;
    HIP_TRY(c, hipMalloc(&dob.p, n * 4)); HIP_TRY(c, hipMalloc(&del.p, n * 4)); HIP_TRY(c, hipMalloc(&dpos.p, n * 12)); HIP_TRY(c, hipMalloc(&dout.p, n * 12));
    HIP_TRY(c, hipMemcpyAsync(dob.p, objects, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(del.p, elems, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dpos.p, positions, n * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_surface_normals, dim3(batch_blocks(c, n)), dim3(kBlock), small_lds_bytes(c), c->stream, c->view, n, dob.p, del.p, dpos.p, dout.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_normals, dout.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_random_surface_points(mcrt_ctx* c, uint32_t object_index, uint64_t n, uint64_t* rng_states, float* out_points)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (object_index >= c->view.nobj) return fail(c, MCRT_ERR_INVALID, "object index outside the scene");
    if (n == 0) return MCRT_OK;
    if (!rng_states || !out_points) return fail(c, MCRT_ERR_INVALID, "null buffer");
    DevBuf<uint64_t> dst;
    DevBuf<float> dout;
    HIP_TRY(c, hipMalloc(&dst.p, n * 16)); HIP_TRY(c, hipMalloc(&dout.p, n * 12));
    HIP_TRY(c, hipMemcpyAsync(dst.p, rng_states, n * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_surface_points, dim3(batch_blocks(c, n)), dim3(kBlock), small_lds_bytes(c), c->stream, c->view, object_index, n, dst.p, dout.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_points, dout.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rng_states, dst.p, n * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_brdf(mcrt_ctx* c, const mcrt_material* material, uint64_t n, const float* incident, const float* reflection, const float* normal, double* out)
{
    if (!c) return MCRT_ERR_INVALID;
    int rc = bind_device(c);
    if (rc) return rc;
    if (!material) return fail(c, MCRT_ERR_INVALID, "null material");
    if (material->diffuse < 0 || material->diffuse > 1) return fail(c, MCRT_ERR_INVALID, "material enum out of range");
    if (n == 0) return MCRT_OK;
    if (!incident || !reflection || !normal || !out) return fail(c, MCRT_ERR_INVALID, "null buffer");
    float on_a, on_b;
    oren_nayar_terms(material->roughness, on_a, on_b);
    DevBuf<float> di, dr, dn;
    DevBuf<double> dout;
    HIP_TRY(c, hipMalloc(&di.p, n * 12)); HIP_TRY(c, hipMalloc(&dr.p, n * 12)); HIP_TRY(c, hipMalloc(&dn.p, n * 12)); HIP_TRY(c, hipMalloc(&dout.p, n * 8));
    HIP_TRY(c, hipMemcpyAsync(di.p, incident, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dr.p, reflection, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dn.p, normal, n * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_brdf, dim3(batch_blocks(c, n)), dim3(kBlock), 0, c->stream, (uint32_t)material->diffuse, material->albedo, on_a, on_b, n, di.p, dr.p, dn.p, dout.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_trace_samples(mcrt_ctx* c, uint64_t n, const uint32_t* pixels, const uint32_t* samples, uint64_t seed, uint32_t max_depth, double* out)
{
    int rc = need_scene(c, true);
    if (rc) return rc;
    if (n == 0) return MCRT_OK;
    if (!pixels || !samples || !out) return fail(c, MCRT_ERR_INVALID, "null buffer");
    if (max_depth > kMaxDepthLimit) return fail(c, MCRT_ERR_LIMIT, "max_depth above 256");
    const uint64_t npix = (uint64_t)c->cam.width * c->cam.height;
    for (uint64_t i = 0; i < n; ++i)
        if (pixels[i] >= npix) return fail(c, MCRT_ERR_INVALID, "pixel index outside the image");
    rc = ensure_pend(c, max_depth);
    if (rc) return rc;
    DevBuf<uint32_t> dp, ds;
    DevBuf<double> dout;
    HIP_TRY(c, hipMalloc(&dp.p, n * 4)); HIP_TRY(c, hipMalloc(&ds.p, n * 4)); HIP_TRY(c, hipMalloc(&dout.p, n * 24));
    HIP_TRY(c, hipMemcpyAsync(dp.p, pixels, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ds.p, samples, n * 4, hipMemcpyHostToDevice, c->stream));
    if (c->counting) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
    ListArgs a;
    a.view = view_for(c); a.cam = c->cam; a.pixels = dp.p; a.samples = ds.p; a.out = dout.p; a.counters = c->d_counters; a.pend = c->d_pend;
    a.n = n; a.max_depth = max_depth; a.seed = seed;
    a.ray_o = nullptr; a.ray_d = nullptr; a.ray_state = nullptr; a.ray_importance = nullptr;
    a.refill = lane_policy(c).refill;
    rc = launch_list(c, a);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, n * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_trace_rays(mcrt_ctx* c, uint64_t n, const float* origins, const float* dirs, uint64_t* rng_states, uint32_t max_depth,
                    const double* importance, double* out)
{
    int rc = need_scene(c, false);
    if (rc) return rc;
    if (n == 0) return MCRT_OK;
    if (!origins || !dirs || !rng_states || !out) return fail(c, MCRT_ERR_INVALID, "null buffer");
    if (max_depth > kMaxDepthLimit) return fail(c, MCRT_ERR_LIMIT, "max_depth above 256");
    rc = ensure_pend(c, max_depth);
    if (rc) return rc;
    DevBuf<float> dor, ddi;
    DevBuf<uint64_t> dst;
    DevBuf<double> dimp, dout;
    HIP_TRY(c, hipMalloc(&dor.p, n * 12)); HIP_TRY(c, hipMalloc(&ddi.p, n * 12)); HIP_TRY(c, hipMalloc(&dst.p, n * 16));
    HIP_TRY(c, hipMalloc(&dout.p, n * 24));
    HIP_TRY(c, hipMemcpyAsync(dor.p, origins, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ddi.p, dirs, n * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dst.p, rng_states, n * 16, hipMemcpyHostToDevice, c->stream));
    if (importance) {
        HIP_TRY(c, hipMalloc(&dimp.p, n * 24));
        HIP_TRY(c, hipMemcpyAsync(dimp.p, importance, n * 24, hipMemcpyHostToDevice, c->stream));
    }
    if (c->counting) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
    ListArgs a;
    a.view = view_for(c); a.cam = c->cam; a.pixels = nullptr; a.samples = nullptr; a.out = dout.p; a.counters = c->d_counters; a.pend = c->d_pend;
    a.n = n; a.max_depth = max_depth; a.seed = 0;
    a.ray_o = dor.p; a.ray_d = ddi.p; a.ray_state = dst.p; a.ray_importance = dimp.p;
    a.refill = lane_policy(c).refill;
    rc = launch_list(c, a);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, n * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rng_states, dst.p, n * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_render_unlit(mcrt_ctx* c, double* out)
{
    int rc = need_scene(c, true);
    if (rc) return rc;
    if (!out) return fail(c, MCRT_ERR_INVALID, "null buffer");
    const size_t bytes = c->accum_pixels * 3 * sizeof(double);
    DevBuf<double> d;
    HIP_TRY(c, hipMalloc(&d.p, bytes));
    rc = launch_unlit(c, d.p);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, d.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

void mcrt_rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t out_state[2])
{
    uint64_t a, b;
    sample_seed(seed, pixel, sample, a, b);
    out_state[0] = a; out_state[1] = b;
}

int mcrt_clear(mcrt_ctx* c)
{
    if (!c) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    int rc = bind_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, c->accum_pixels * 3 * sizeof(double), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_dropped, 0, sizeof(unsigned long long), c->stream));
    return MCRT_OK;
}

int mcrt_dropped_samples(mcrt_ctx* c, uint64_t* out)
{
    if (!c || !out) return MCRT_ERR_INVALID;
    int rc = bind_device(c);
    if (rc) return rc;
    unsigned long long n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, c->d_dropped, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out = n;
    return MCRT_OK;
}

int mcrt_bind_accumulator(mcrt_ctx* c, double* d_accum)
{
    if (!c) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    c->d_accum = d_accum ? d_accum : c->d_accum_own;
    return MCRT_OK;
}

int mcrt_accumulator(mcrt_ctx* c, double** out)
{
    if (!c || !out) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    *out = c->d_accum;
    return MCRT_OK;
}

int mcrt_render(mcrt_ctx* c, uint32_t sample_begin, uint32_t sample_count, uint32_t max_depth, uint64_t seed)
{
    int rc = need_scene(c, true);
    if (rc) return rc;
    if (max_depth > kMaxDepthLimit) return fail(c, MCRT_ERR_LIMIT, "max_depth above 256");
    if (sample_count == 0) return MCRT_OK;
    const bool wavefront = use_wavefront(c);
    rc = wavefront ? ensure_wavefront(c, max_depth) : ensure_pend(c, max_depth);
    if (rc) return rc;
    const bool parked = !wavefront && c->accel == MCRT_ACCEL_OCTREE && !c->big_in_lds;
    if (parked && (rc = ensure_park(c)) != MCRT_OK) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_tile, 0, 2 * sizeof(uint32_t), c->stream));
    if (c->counting) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
    RenderArgs a;
    a.view = view_for(c); a.cam = c->cam; a.accum = c->d_accum; a.tile_counter = c->d_tile; a.dropped = c->d_dropped; a.counters = c->d_counters; a.pend = c->d_pend;
    a.park = parked ? c->d_park : nullptr;
    a.s_begin = sample_begin; a.s_count = sample_count; a.max_depth = max_depth;
    a.policy = lane_policy(c);
    a.slices = c->lanes_override ? std::min<uint32_t>(c->lanes_override, sample_count) : mcrt_lanes_per_pixel(c->cam.width, c->cam.height, sample_count);
    {
        // the slices of a pixel share a wavefront (4 slices: 4x4 pixels x 4 slices): a smaller pixel footprint per
        // wavefront keeps its primary rays, first hits and shading branches together (+6..8 % measured);
        // MCRT_WAVE_SLICES overrides (tuning knob)
        const uint32_t ws = c->wave_slices;
        uint32_t lg = 0;
        while ((2u << lg) <= ws && (2u << lg) <= a.slices && a.slices % (2u << lg) == 0 && lg < 6) ++lg;
        a.wave_slices_log2 = lg;
        const uint32_t log_p = 6u - lg, tw = 1u << ((log_p + 1u) >> 1), th = 1u << (log_p >> 1);
        a.tiles_x = (c->cam.width + tw - 1u) / tw;
        a.tiles = a.tiles_x * ((c->cam.height + th - 1u) / th);
    }
    if ((uint64_t)a.tiles * 64u * (a.slices >> a.wave_slices_log2) > 0xFFFF0000ull) return fail(c, MCRT_ERR_LIMIT, "image too large for one pass");
    a.items = a.tiles * 64u * (a.slices >> a.wave_slices_log2);
    a.partial = nullptr;
    if (a.slices > 1) {
        const size_t need = (size_t)a.slices * c->accum_pixels * 24;
        if (need > c->partial_bytes) {
            if (c->d_partial) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_partial); c->d_partial = nullptr; c->partial_bytes = 0; }
            HIP_TRY(c, hipMalloc(&c->d_partial, need));
            c->partial_bytes = need;
        }
        a.partial = c->d_partial;
    }
    a.seed = seed;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    rc = wavefront ? render_wavefront(c, a) : launch_render(c, a);
    if (rc) return rc;
    if (a.slices > 1) {
        const uint64_t values = (uint64_t)c->accum_pixels * 3;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((values + kBlock - 1) / kBlock, 65535);
        hipLaunchKernelGGL(k_combine, dim3(blocks), dim3(kBlock), 0, c->stream, c->d_accum, (const double*)c->d_partial, values, a.slices);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return MCRT_OK;
}

int mcrt_add_accumulator(mcrt_ctx* c, const double* d_src)
{
    if (!c || !d_src) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    int rc = bind_device(c);
    if (rc) return rc;
    const uint64_t values = (uint64_t)c->accum_pixels * 3;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((values + kBlock - 1) / kBlock, 65535);
    hipLaunchKernelGGL(k_add, dim3(blocks), dim3(kBlock), 0, c->stream, c->d_accum, d_src, values);
    HIP_TRY(c, hipGetLastError());
    return MCRT_OK;
}

int mcrt_comm_init_all(int n, const int* devices, void** out_comms)
{
    if (n <= 0 || !devices || !out_comms) return fail(nullptr, MCRT_ERR_INVALID, "mcrt_comm_init_all: bad arguments");
    const Rccl& r = rccl();
    if (!r.ok) return fail(nullptr, MCRT_ERR_NO_DEVICE, r.why);
    const int e = r.init_all(out_comms, n, devices);
    if (e != 0) return fail(nullptr, MCRT_ERR_HIP, std::string("ncclCommInitAll: ") + (r.err ? r.err(e) : "error"));
    return MCRT_OK;
}

int mcrt_comm_destroy(void* comm)
{
    if (!comm) return MCRT_OK;
    const Rccl& r = rccl();
    if (!r.ok) return fail(nullptr, MCRT_ERR_NO_DEVICE, r.why);
    return r.destroy(comm) == 0 ? MCRT_OK : fail(nullptr, MCRT_ERR_HIP, "ncclCommDestroy failed");
}

int mcrt_reduce(mcrt_ctx* c, void* comm, int root)
{
    if (!c || !comm) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    const Rccl& r = rccl();
    if (!r.ok) return fail(c, MCRT_ERR_NO_DEVICE, r.why);
    int rc = bind_device(c);
    if (rc) return rc;
    // one collective per pass: the W*H*3 doubles of every rank summed into `root`, in place, on the context's stream
    const int e = r.reduce(c->d_accum, c->d_accum, c->accum_pixels * 3, kNcclDouble, kNcclSum, root, comm, c->stream);
    if (e != 0) return fail(c, MCRT_ERR_HIP, std::string("ncclReduce: ") + (r.err ? r.err(e) : "error"));
    return MCRT_OK;
}

uint32_t mcrt_lanes_per_pixel(uint32_t width, uint32_t height, uint32_t sample_count)
{
    // smallest power of two that brings pixels * slices to 2^22 work items (16 tile-sized fetches per resident
    // wavefront: measured, coarser items leave the GPU half empty while the last tiles finish), never more slices
    // than an eighth of the samples (items of fewer than 8 samples cost more in fetch/flush than they balance)
    const uint64_t pixels = (uint64_t)width * height;
    uint32_t l = 1;
    while (l < 64u && pixels * l < (1ull << 22) && l * 16u <= sample_count) l *= 2u;
    return l;
}

int mcrt_set_lanes_per_pixel(mcrt_ctx* c, uint32_t lanes)
{
    if (!c) return MCRT_ERR_INVALID;
    if (lanes > 64u) return fail(c, MCRT_ERR_LIMIT, "at most 64 sample slices per pixel");
    c->lanes_override = lanes;
    return MCRT_OK;
}

int mcrt_last_render_ms(mcrt_ctx* c, float* out_ms)
{
    if (!c || !out_ms) return MCRT_ERR_INVALID;
    if (!c->timed) return fail(c, MCRT_ERR_STATE, "no render launched yet");
    int rc = bind_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(out_ms, c->ev0, c->ev1));
    return MCRT_OK;
}

int mcrt_resolve(mcrt_ctx* c, int tonemap, double gamma, double exposure, uint64_t spp_total, uint8_t* out_bgra8, double* out_mean)
{
    if (!c) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    if (tonemap < MCRT_TONEMAP_NONE || tonemap > MCRT_TONEMAP_EXPOSURE) return fail(c, MCRT_ERR_INVALID, "unknown tonemap");
    if (spp_total == 0) return fail(c, MCRT_ERR_INVALID, "spp_total is zero");
    int rc = bind_device(c);
    if (rc) return rc;
    const uint64_t npix = c->accum_pixels;
    // the staging images live with the context (a progressive driver resolves after every pass)
    if (out_bgra8 && c->resolve_img_pixels < npix) {
        if (c->d_resolve_img) { (void)hipFree(c->d_resolve_img); c->d_resolve_img = nullptr; c->resolve_img_pixels = 0; }
        HIP_TRY(c, hipMalloc(&c->d_resolve_img, npix * 4));
        c->resolve_img_pixels = npix;
    }
    if (out_mean && c->resolve_mean_pixels < npix) {
        if (c->d_resolve_mean) { (void)hipFree(c->d_resolve_mean); c->d_resolve_mean = nullptr; c->resolve_mean_pixels = 0; }
        HIP_TRY(c, hipMalloc(&c->d_resolve_mean, npix * 24));
        c->resolve_mean_pixels = npix;
    }
    uint8_t* dimg = out_bgra8 ? c->d_resolve_img : nullptr;
    double* dmean = out_mean ? c->d_resolve_mean : nullptr;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((npix + kBlock - 1) / kBlock, 65535);
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(kBlock), 0, c->stream, (const double*)c->d_accum, npix, (double)spp_total, tonemap,
                       1.0 / gamma, exposure, dimg, dmean);
    HIP_TRY(c, hipGetLastError());
    if (out_bgra8) HIP_TRY(c, hipMemcpyAsync(out_bgra8, dimg, npix * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_mean) HIP_TRY(c, hipMemcpyAsync(out_mean, dmean, npix * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_read_accumulator(mcrt_ctx* c, double* out)
{
    if (!c || !out) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    int rc = bind_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->d_accum, c->accum_pixels * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_write_accumulator(mcrt_ctx* c, const double* sums)
{
    if (!c || !sums) return MCRT_ERR_INVALID;
    if (!c->have_camera) return fail(c, MCRT_ERR_STATE, "no camera set (mcrt_set_camera)");
    int rc = bind_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_accum, sums, c->accum_pixels * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_synchronize(mcrt_ctx* c)
{
    if (!c) return MCRT_ERR_INVALID;
    int rc = bind_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

int mcrt_octree_info_get(mcrt_ctx* c, mcrt_octree_info* out)
{
    if (!c || !out) return MCRT_ERR_INVALID;
    if (!c->have_scene) return fail(c, MCRT_ERR_STATE, "no scene uploaded (mcrt_upload_scene)");
    out->node_count = c->view.nnode;
    out->prim_ref_count = c->flat.ref_count;
    out->prim_count = c->view.nprim;
    out->max_depth = c->flat.tree_depth;
    out->blob_bytes = (c->view.small_words + c->view.big_words) * 4;
    out->lds_resident = c->big_in_lds ? 1u : 0u;
    for (int a = 0; a < 3; ++a) { out->root_cell_min[a] = c->flat.root_cell[a]; out->root_cell_max[a] = c->flat.root_cell[3 + a]; }
    out->pad = c->flat.pad;
    out->far_origin_limit = c->view.far_limit;
    return MCRT_OK;
}

int mcrt_octree_download(mcrt_ctx* c, float* node_box, uint32_t* node_words, uint32_t* prim_refs)
{
    if (!c) return MCRT_ERR_INVALID;
    if (!c->have_scene) return fail(c, MCRT_ERR_STATE, "no scene uploaded (mcrt_upload_scene)");
    int rc = ensure_tree_on_host(c);
    if (rc) return rc;
    if (node_box) std::memcpy(node_box, c->flat.node_box.data(), c->flat.node_box.size() * 4);
    if (node_words) std::memcpy(node_words, c->flat.node_words.data(), c->flat.node_words.size() * 4);
    if (prim_refs) std::memcpy(prim_refs, c->flat.refs.data(), c->flat.refs.size() * 4);
    return MCRT_OK;
}

int mcrt_scene_image_create(const mcrt_scene_desc* scene, const mcrt_accel_params* accel, mcrt_scene_image** out_image)
{
    if (!out_image) return fail(nullptr, MCRT_ERR_INVALID, "out_image is null");
    *out_image = nullptr;
    FlatScene* f = new (std::nothrow) FlatScene;
    if (!f) return fail(nullptr, MCRT_ERR_INVALID, "out of host memory");
    std::string err;
    int rc = flatten_scene(scene, accel, *f, err);
    if (rc) { delete f; return fail(nullptr, rc, err); }
    *out_image = reinterpret_cast<mcrt_scene_image*>(f);
    return MCRT_OK;
}

int mcrt_scene_image_destroy(mcrt_scene_image* image)
{
    delete reinterpret_cast<FlatScene*>(image);
    return MCRT_OK;
}

int mcrt_scene_image_info(const mcrt_scene_image* image, mcrt_octree_info* out)
{
    if (!image || !out) return fail(nullptr, MCRT_ERR_INVALID, "null argument");
    const FlatScene& f = *reinterpret_cast<const FlatScene*>(image);
    out->node_count = f.view.nnode;
    out->prim_ref_count = f.ref_count;
    out->prim_count = f.view.nprim;
    out->max_depth = f.tree_depth;
    out->blob_bytes = (f.view.small_words + f.view.big_words) * 4;
    const size_t small_bytes = (((size_t)f.view.small_words + 3u) & ~size_t(3)) * 4;
    out->lds_resident = (small_bytes + (size_t)f.view.big_words * 4 <= kLdsBudgetBytes) ? 1u : 0u;
    for (int a = 0; a < 3; ++a) { out->root_cell_min[a] = f.root_cell[a]; out->root_cell_max[a] = f.root_cell[3 + a]; }
    out->pad = f.pad;
    out->far_origin_limit = f.view.far_limit;
    return MCRT_OK;
}

int mcrt_scene_image_octree(const mcrt_scene_image* image, float* node_box, uint32_t* node_words, uint32_t* prim_refs)
{
    if (!image) return fail(nullptr, MCRT_ERR_INVALID, "null argument");
    const FlatScene& f = *reinterpret_cast<const FlatScene*>(image);
    if (node_box) std::memcpy(node_box, f.node_box.data(), f.node_box.size() * 4);
    if (node_words) std::memcpy(node_words, f.node_words.data(), f.node_words.size() * 4);
    if (prim_refs) std::memcpy(prim_refs, f.refs.data(), f.refs.size() * 4);
    return MCRT_OK;
}

int mcrt_set_counting(mcrt_ctx* c, int enabled)
{
    if (!c) return MCRT_ERR_INVALID;
    c->counting = enabled != 0;
    return MCRT_OK;
}

int mcrt_counters(mcrt_ctx* c, uint64_t out[8])
{
    if (!c || !out) return MCRT_ERR_INVALID;
    int rc = bind_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long tmp[8];
    HIP_TRY(c, hipMemcpy(tmp, c->d_counters, sizeof tmp, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out[i] = tmp[i];
    return MCRT_OK;
}

} // extern "C"
