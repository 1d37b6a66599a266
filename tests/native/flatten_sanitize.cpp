// Host-only check of the scene flattener (octree build, image layout) under AddressSanitizer and UBSan.
// Built and run by tests/test_host_and_abi.py::test_flatten_under_sanitizers with plain g++ (no GPU, no HIP).
// Scenes: the Cornell walls, degenerate inputs (zero-area triangles, coincident primitives, a sphere of
// radius below FLT_EPSILON), a deep tree (leaf size 1), and a 200k-triangle soup.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mcrt_flatten.h"

using namespace mcrt;

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float frand(uint32_t& s, float lo, float hi) { return lo + (hi - lo) * (float)lcg(s) * (1.0f / 16777216.0f); }

static mcrt_triangle tri(const float a[3], const float b[3], const float c[3])
{
    mcrt_triangle t{};
    for (int k = 0; k < 3; ++k) { t.v0[k] = a[k]; t.v1[k] = b[k]; t.v2[k] = c[k]; }
    t.normal[0] = 0; t.normal[1] = 1; t.normal[2] = 0;
    return t;
}

static int check(const char* what, const std::vector<mcrt_object>& objs, const std::vector<mcrt_triangle>& tris, const mcrt_accel_params* ap, int want_rc)
{
    mcrt_scene_desc d{};
    d.object_count = (uint32_t)objs.size(); d.triangle_count = (uint32_t)tris.size();
    d.objects = objs.data(); d.triangles = tris.data();
    FlatScene out; std::string err;
    int rc = flatten_scene(&d, ap, out, err);
    if (rc != want_rc) { std::printf("FAIL %s: rc %d (%s), expected %d\n", what, rc, err.c_str(), want_rc); return 1; }
    if (rc == MCRT_OK) {
        const SceneView& v = out.view;
        // every child index and leaf index of every node stays inside its array; every list is terminated
        for (uint32_t n = 0; n < v.nnode; ++n) {
            const uint32_t* w = out.big_img.data() + v.off_node + (size_t)n * NODE_STRIDE;
            const uint32_t mask = w[NODE_CHILD] >> 24, base = w[NODE_CHILD] & 0xFFFFFFu;
            if (mask && base + (uint32_t)__builtin_popcount(mask) > v.nnode) { std::printf("FAIL %s: child range\n", what); return 1; }
            if (w[NODE_LEAF] != NO_REFS) {
                uint32_t r = w[NODE_LEAF];
                for (;;) {
                    if (r >= out.ref_count) { std::printf("FAIL %s: unterminated leaf list\n", what); return 1; }
                    const uint32_t* rec = out.big_img.data() + v.off_leaf + (size_t)r * LEAF_STRIDE;
                    if ((rec[LEAF_PRIM] & ~REF_LAST) >= v.nprim) { std::printf("FAIL %s: primitive id\n", what); return 1; }
                    if (rec[LEAF_PRIM] & REF_LAST) break;
                    ++r;
                }
            }
        }
        std::printf("ok   %s: %u nodes, %u leaf records, depth %u, %.1f MB\n", what, v.nnode, out.ref_count, out.tree_depth, v.big_words * 4e-6);
    } else std::printf("ok   %s: refused (%s)\n", what, err.c_str());
    return 0;
}

int main()
{
    int bad = 0;
    uint32_t seed = 12345;
    auto mesh = [](uint32_t first, uint32_t count) {
        mcrt_object o{};
        o.kind = MCRT_OBJECT_MESH; o.first_triangle = first; o.triangle_count = count;
        for (int k = 0; k < 3; ++k) { o.aabb_min[k] = -5; o.aabb_max[k] = 5; }
        o.material.color[0] = o.material.color[1] = o.material.color[2] = 0.5; o.material.albedo = 1; o.material.refractive_index = 1.5f;
        return o;
    };
    auto sphere = [](float x, float y, float z, float r) {
        mcrt_object o{};
        o.kind = MCRT_OBJECT_SPHERE; o.radius = r; o.position[0] = x; o.position[1] = y; o.position[2] = z;
        o.material.albedo = 1; o.material.refractive_index = 1.5f;
        return o;
    };
    {   // soup + spheres, default parameters and a deep tree
        std::vector<mcrt_triangle> tris;
        for (int i = 0; i < 200000; ++i) {
            float c[3] = { frand(seed, -4, 4), frand(seed, -4, 4), frand(seed, -4, 4) }, a[3], b[3], d[3];
            for (int k = 0; k < 3; ++k) { a[k] = c[k] + frand(seed, -.05f, .05f); b[k] = c[k] + frand(seed, -.05f, .05f); d[k] = c[k] + frand(seed, -.05f, .05f); }
            tris.push_back(tri(a, b, d));
        }
        std::vector<mcrt_object> objs = { mesh(0, 150000), sphere(0, 0, 0, 1.0f), mesh(150000, 50000), sphere(1, 1, 1, 1e-9f) };
        bad += check("soup 200k", objs, tris, nullptr, MCRT_OK);
        mcrt_accel_params deep = { 1, 12, 0.0f };
        std::vector<mcrt_triangle> few(tris.begin(), tris.begin() + 3000);
        std::vector<mcrt_object> o2 = { mesh(0, 3000) };
        bad += check("deep tree", o2, few, &deep, MCRT_OK);
    }
    {   // degenerate: identical and zero-area triangles (a split never separates them)
        float p[3] = { 1, 1, 1 }, q[3] = { 1, 1, 1 }, r[3] = { 2, 1, 1 };
        std::vector<mcrt_triangle> tris(500, tri(p, q, r));
        std::vector<mcrt_object> objs = { mesh(0, 500) };
        bad += check("coincident", objs, tris, nullptr, MCRT_OK);
    }
    {   // invalid inputs are refused, not read out of bounds
        float p[3] = { 0, 0, 0 }, q[3] = { 1, 0, 0 }, r[3] = { 0, 1, 0 };
        std::vector<mcrt_triangle> tris(4, tri(p, q, r));
        std::vector<mcrt_object> objs = { mesh(2, 10) };
        bad += check("range outside", objs, tris, nullptr, MCRT_ERR_INVALID);
        objs = { mesh(0, 4) }; tris[1].v1[0] = NAN;
        bad += check("nan vertex", objs, tris, nullptr, MCRT_ERR_INVALID);
        std::vector<mcrt_object> many(513, sphere(0, 0, 0, 1));
        bad += check("513 objects", many, {}, nullptr, MCRT_ERR_LIMIT);
        bad += check("empty", {}, {}, nullptr, MCRT_ERR_INVALID);
    }
    std::printf(bad ? "FAILED\n" : "all ok\n");
    return bad ? 1 : 0;
}
