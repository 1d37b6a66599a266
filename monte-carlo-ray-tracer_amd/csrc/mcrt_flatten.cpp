// mcrt_flatten.cpp - host half of mcrt_upload_scene.
//
// Replaces Scene::PrepareForRayTracing (source/scene.cpp:38-59): the light
// cache (:41-49) becomes an index list, and Octree::Fill / Subdivide /
// InsertIfOverlaps (source/accelerationstructures/octree.h:50-92,168-223) are
// rebuilt as a per-PRIMITIVE octree (the reference's is per object, never
// queried and non-functional) that is then flattened
// breadth-first into the node / leaf records of mcrt_layout.h.
//
// What is kept from the reference structure: 8-way axis-aligned subdivision
// at the cell centre, a `maxCount` leaf threshold (octree.h:31), the root box
// as the union of the member boxes (octree.h:83-86), and the "give up when a
// split separates nothing" rule (octree.h:195-212).  What differs, so that
// queries equal the brute force of source/scene.cpp:62-80 exactly:
//   - members are single triangles / spheres with their TRUE bounds (the
//     reference boxes spheres at half size, source/objects/sphere.h:71-74);
//   - a primitive much larger than a child cell stays in the interior node's
//     own list; the others go into every child whose cell the triangle itself
//     (not just its box) overlaps;
//   - every node stores the union of (padded member box  intersect  padded
//     cell), so a float slab test on it can never cull a node holding the
//     nearest hit - also not for a ray lying exactly in a split plane.
#include "mcrt_flatten.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace mcrt {
namespace {

struct Box {
    float mn[3], mx[3];
};

inline Box empty_box()
{
    Box b;
    for (int a = 0; a < 3; ++a) { b.mn[a] = FLT_MAX; b.mx[a] = -FLT_MAX; }
    return b;
}
inline void grow(Box& b, const Box& o)
{
    for (int a = 0; a < 3; ++a) { b.mn[a] = std::min(b.mn[a], o.mn[a]); b.mx[a] = std::max(b.mx[a], o.mx[a]); }
}
inline bool overlaps(const Box& a, const Box& b)
{
    for (int k = 0; k < 3; ++k) if (a.mx[k] < b.mn[k] || b.mx[k] < a.mn[k]) return false;
    return true;
}
inline Box clip(const Box& a, const Box& cell)
{
    Box r;
    for (int k = 0; k < 3; ++k) { r.mn[k] = std::max(a.mn[k], cell.mn[k]); r.mx[k] = std::min(a.mx[k], cell.mx[k]); }
    return r;
}

// Triangle / box overlap by separating axes (Akenine-Moller 2001), in double.  `box` is already expanded by the
// padding, so "overlaps" errs on the side of keeping a reference.  A primitive's padded AABB overlapping a cell
// is necessary but far from sufficient for a thin triangle: on a triangle soup the AABB test alone keeps 1.5x
// the references (and triangle tests per ray) that this test keeps.
struct TriD { double v[3][3]; };

inline bool tri_overlaps_box(const TriD& t, const Box& box)
{
    double c[3], h[3], v[3][3];
    for (int a = 0; a < 3; ++a) { c[a] = 0.5 * ((double)box.mn[a] + (double)box.mx[a]); h[a] = 0.5 * ((double)box.mx[a] - (double)box.mn[a]); }
    for (int i = 0; i < 3; ++i) for (int a = 0; a < 3; ++a) v[i][a] = t.v[i][a] - c[a];
    const double e[3][3] = { { v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2] },
                             { v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2] },
                             { v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2] } };
    const double eps = 1e-12;
    // nine cross-product axes: (box axis a) x (edge i)
    for (int i = 0; i < 3; ++i)
        for (int a = 0; a < 3; ++a) {
            const int b = (a + 1) % 3, cc = (a + 2) % 3;
            // axis = unit(a) x e[i] = (0, -e[cc], e[b]) in (a, b, cc) order
            const double ax_b = -e[i][cc], ax_c = e[i][b];
            double lo = 1e300, hi = -1e300;
            for (int k = 0; k < 3; ++k) { const double p = ax_b * v[k][b] + ax_c * v[k][cc]; lo = std::min(lo, p); hi = std::max(hi, p); }
            const double r = h[b] * std::fabs(ax_b) + h[cc] * std::fabs(ax_c) + eps;
            if (lo > r || hi < -r) return false;
        }
    // the three box axes
    for (int a = 0; a < 3; ++a) {
        const double lo = std::min(v[0][a], std::min(v[1][a], v[2][a])), hi = std::max(v[0][a], std::max(v[1][a], v[2][a]));
        if (lo > h[a] + eps || hi < -h[a] - eps) return false;
    }
    // the triangle's plane
    const double n[3] = { e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0] };
    const double d = n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2];
    const double r = h[0] * std::fabs(n[0]) + h[1] * std::fabs(n[1]) + h[2] * std::fabs(n[2]) + eps;
    return std::fabs(d) <= r;
}

struct BuildNode {
    Box cell;                 // the octant this node owns
    Box tight;                // union of member boxes clipped to the cell (whole subtree)
    std::vector<uint32_t> own;  // primitives tested when the node is visited
    int child[8];             // index into the build-node pool or -1
    uint32_t depth;
};

struct Builder {
    const std::vector<Box>& prim_box;   // padded primitive bounds
    const std::vector<TriD>& prim_tri;  // the triangle of each primitive (unused entries for spheres)
    const std::vector<uint8_t>& is_tri;
    float pad;
    float stay;
    uint32_t max_leaf, max_depth;
    std::vector<BuildNode> pool;
    uint32_t deepest = 0;

    int make(const Box& cell, std::vector<uint32_t>& members, uint32_t depth)
    {
        int id = int(pool.size());
        pool.emplace_back();
        {
            BuildNode& n = pool.back();
            n.cell = cell;
            n.depth = depth;
            for (int c = 0; c < 8; ++c) n.child[c] = -1;
            // clipped to the PADDED cell: a ray that runs exactly in a split plane (zero direction component, origin on
            // the plane - every primary ray through the image centre column of a symmetric room) must find both
            // neighbours' boxes open on that axis, and a hit point a rounding error away from the plane must lie
            // strictly inside the box of the cell that holds it
            Box padded_cell;
            for (int a = 0; a < 3; ++a) { padded_cell.mn[a] = cell.mn[a] - pad; padded_cell.mx[a] = cell.mx[a] + pad; }
            n.tight = empty_box();
            for (uint32_t p : members) grow(n.tight, clip(prim_box[p], padded_cell));
        }
        deepest = std::max(deepest, depth);

        if (members.size() <= max_leaf || depth >= max_depth) {
            pool[id].own.swap(members);
            return id;
        }

        float half[3], centre[3];
        for (int a = 0; a < 3; ++a) { half[a] = 0.5f * (cell.mx[a] - cell.mn[a]); centre[a] = cell.mn[a] + half[a]; }
        const float child_extent = std::max(half[0], std::max(half[1], half[2]));

        // Large primitives stay here; the rest go down.  "Large" = wider than `stay` x the child extent.  stay = 4
        // (more than two cells wide) for big scenes: anything smaller reaches at most a few dozen cells of its own
        // size class, so references stay bounded, while keeping half-cell-sized primitives up here (stay = 1) made
        // every ray that crossed the cell test them: 283 instead of 111 triangle tests per sample on the 1M-triangle
        // soup.  stay = 1 for scenes small enough to live in LDS, where image size decides occupancy.
        const float kStay = stay;
        std::vector<uint32_t> own, down;
        for (uint32_t p : members) {
            const Box& b = prim_box[p];
            float ext = std::max(b.mx[0] - b.mn[0], std::max(b.mx[1] - b.mn[1], b.mx[2] - b.mn[2]));
            (ext > kStay * child_extent ? own : down).push_back(p);
        }
        members.clear();
        members.shrink_to_fit();

        if (!down.empty()) {
            Box cells[8];
            std::vector<uint32_t> lists[8];
            for (int c = 0; c < 8; ++c)
                for (int a = 0; a < 3; ++a) {
                    bool hi = (c >> a) & 1;
                    cells[c].mn[a] = hi ? centre[a] : cell.mn[a];
                    cells[c].mx[a] = hi ? cell.mx[a] : centre[a];
                }
            Box padded[8];
            for (int c = 0; c < 8; ++c)
                for (int a = 0; a < 3; ++a) { padded[c].mn[a] = cells[c].mn[a] - pad; padded[c].mx[a] = cells[c].mx[a] + pad; }
            size_t placed = 0;
            for (uint32_t p : down)
                for (int c = 0; c < 8; ++c)
                    if (overlaps(prim_box[p], cells[c]) && (!is_tri[p] || tri_overlaps_box(prim_tri[p], padded[c]))) { lists[c].push_back(p); ++placed; }
            // The reference's guard (octree.h:195-212): a split that repeats every member in
            // every child separates nothing.
            if (placed >= down.size() * 8) {
                own.insert(own.end(), down.begin(), down.end());
            } else {
                down.clear();
                down.shrink_to_fit();
                for (int c = 0; c < 8; ++c)
                    if (!lists[c].empty()) {
                        int ch = make(cells[c], lists[c], depth + 1);
                        pool[id].child[c] = ch;
                    }
            }
        }
        std::sort(own.begin(), own.end());
        pool[id].own.swap(own);
        return id;
    }
};

inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

void put_double(std::vector<uint32_t>& img, size_t at, double v)
{
    uint64_t u; std::memcpy(&u, &v, 8);
    img[at] = uint32_t(u & 0xFFFFFFFFu);
    img[at + 1] = uint32_t(u >> 32);
}

bool finite3(const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

} // namespace

namespace {
struct F3 { float x, y, z; };
inline F3 sub3(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot3(F3 a, F3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
inline F3 cross3(F3 x, F3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline F3 norm3(F3 v) { float inv = 1.0f / std::sqrt(dot3(v, v)); return { v.x * inv, v.y * inv, v.z * inv }; }
} // namespace

// A and B of Material::BRDF, source/core/material.h:29-33: functions of the roughness alone (float arithmetic, one rounding per operation)
void oren_nayar_terms(float roughness, float& A, float& B)
{
    volatile float sigma2 = roughness * roughness;
    volatile float qa = 0.5f * sigma2 / (sigma2 + 0.57f);
    volatile float qb = 0.45f * sigma2 / (sigma2 + 0.09f);
    volatile float a = 1.0f - qa;
    A = a; B = qb;
}

void camera_look_at(uint32_t width, uint32_t height, float fov_y_degrees, const float position[3], const float look_at[3],
                    const float up[3], mcrt_camera* out)
{
    // Camera ctor, source/core/camera.h:23-25
    const float halfAngle = fov_y_degrees * 0.5f;
    const float radians = halfAngle / 180.0f * float(3.14159265358979323846264338327950288);
    out->fov_scale = std::tan(radians);
    // glm::lookAtRH, include/glm/ext/matrix_transform.inl:99-119
    const F3 eye{ position[0], position[1], position[2] }, centre{ look_at[0], look_at[1], look_at[2] }, upv{ up[0], up[1], up[2] };
    const F3 f = norm3(sub3(centre, eye));
    const F3 s = norm3(cross3(f, upv));
    const F3 u = cross3(s, f);
    float m[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };   // m[column][row]
    m[0][0] = s.x; m[1][0] = s.y; m[2][0] = s.z;
    m[0][1] = u.x; m[1][1] = u.y; m[2][1] = u.z;
    m[0][2] = -f.x; m[1][2] = -f.y; m[2][2] = -f.z;
    m[3][0] = -dot3(s, eye); m[3][1] = -dot3(u, eye); m[3][2] = dot3(f, eye);
    // glm::inverse(mat4): cofactor expansion, include/glm/detail/func_matrix.inl:294-352
    float c[6][3];   // the six 2x2 sub-determinant triples (Coef00..Coef23 grouped as Fac0..Fac5 uses them)
    auto det2 = [&](int ca, int ra, int cb, int rb) { return m[ca][ra] * m[cb][rb] - m[cb][ra] * m[ca][rb]; };
    // Fac0: rows (2,3); Fac1: rows (1,3); Fac2: rows (1,2); Fac3: rows (0,3); Fac4: rows (0,2); Fac5: rows (0,1);
    // each over column pairs (2,3), (1,3), (1,2)
    const int rows[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
    const int cols[3][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 } };
    for (int k = 0; k < 6; ++k)
        for (int j = 0; j < 3; ++j) c[k][j] = det2(cols[j][0], rows[k][0], cols[j][1], rows[k][1]);
    float inv[4][4];
    const float signA[4] = { +1, -1, +1, -1 }, signB[4] = { -1, +1, -1, +1 };
    for (int i = 0; i < 4; ++i) {
        const int j = i < 2 ? 0 : i - 1;                 // Fac = (Coef_a, Coef_a, Coef_b, Coef_c)
        const int col = i == 0 ? 1 : 0;                  // Vec = (m[1][r], m[0][r], m[0][r], m[0][r])
        const float v0 = m[col][0], v1 = m[col][1], v2 = m[col][2], v3 = m[col][3];
        inv[0][i] = ((v1 * c[0][j] - v2 * c[1][j]) + v3 * c[2][j]) * signA[i];
        inv[1][i] = ((v0 * c[0][j] - v2 * c[3][j]) + v3 * c[4][j]) * signB[i];
        inv[2][i] = ((v0 * c[1][j] - v1 * c[3][j]) + v3 * c[5][j]) * signA[i];
        inv[3][i] = ((v0 * c[2][j] - v1 * c[4][j]) + v2 * c[5][j]) * signB[i];
    }
    const float d0 = m[0][0] * inv[0][0], d1 = m[0][1] * inv[1][0], d2 = m[0][2] * inv[2][0], d3 = m[0][3] * inv[3][0];
    const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) out->view[col * 4 + row] = inv[col][row] * one_over_det;
    out->position[0] = position[0]; out->position[1] = position[1]; out->position[2] = position[2];
    out->width = width; out->height = height;
}

int flatten_scene(const mcrt_scene_desc* sc, const mcrt_accel_params* accel, FlatScene& out, std::string& err, bool build_tree)
{
    if (!sc || (sc->object_count && !sc->objects) || (sc->triangle_count && !sc->triangles)) { err = "null scene description"; return MCRT_ERR_INVALID; }
    if (sc->object_count == 0) { err = "scene has no objects"; return MCRT_ERR_INVALID; }
    if (sc->object_count > MAX_OBJECTS) { err = "more than 512 objects: the object table has to fit in LDS"; return MCRT_ERR_LIMIT; }
    if (sc->triangle_count > 0x3FFFFFFFu) { err = "too many triangles"; return MCRT_ERR_LIMIT; }

    const uint32_t nobj = sc->object_count;
    // ---- validate, number the primitives in (object, element) order ------------------
    std::vector<uint32_t> first_prim(nobj);
    std::vector<uint32_t> prim_obj;
    prim_obj.reserve(size_t(sc->triangle_count) + nobj);
    std::vector<uint32_t> lights;
    for (uint32_t i = 0; i < nobj; ++i) {
        const mcrt_object& o = sc->objects[i];
        first_prim[i] = uint32_t(prim_obj.size());
        if (!finite3(o.position)) { err = "object position is not finite"; return MCRT_ERR_INVALID; }
        if (o.kind == MCRT_OBJECT_SPHERE) {
            if (!std::isfinite(o.radius)) { err = "sphere radius is not finite"; return MCRT_ERR_INVALID; }
            prim_obj.push_back(i);
        } else if (o.kind == MCRT_OBJECT_MESH || o.kind == MCRT_OBJECT_LIGHT) {
            if (uint64_t(o.first_triangle) + o.triangle_count > sc->triangle_count) { err = "mesh triangle range outside the triangle array"; return MCRT_ERR_INVALID; }
            if (!finite3(o.aabb_min) || !finite3(o.aabb_max)) { err = "mesh AABB is not finite"; return MCRT_ERR_INVALID; }
            for (uint32_t t = 0; t < o.triangle_count; ++t) {
                const mcrt_triangle& tr = sc->triangles[o.first_triangle + t];
                if (!finite3(tr.v0) || !finite3(tr.v1) || !finite3(tr.v2)) { err = "triangle vertex is not finite"; return MCRT_ERR_INVALID; }
                prim_obj.push_back(i);
            }
        } else { err = "unknown object kind"; return MCRT_ERR_INVALID; }
        if (o.material.surface < 0 || o.material.surface > 3 || o.material.diffuse < 0 || o.material.diffuse > 1) { err = "material enum out of range"; return MCRT_ERR_INVALID; }
        // light cache, source/scene.cpp:41-49
        if (o.material.emission[0] > 0.0 || o.material.emission[1] > 0.0 || o.material.emission[2] > 0.0) lights.push_back(i);
    }
    const uint32_t nprim = uint32_t(prim_obj.size());
    if (prim_obj.size() > MAX_PRIMITIVES) { err = "more than 8388607 primitives (the length of a node's list is a 23-bit field of its record)"; return MCRT_ERR_LIMIT; }

    out.tree_built = build_tree;
    // default leaf size: 4 for scenes that live in LDS; 8 for the big ones, whose leaf records are dealt over the wavefront (a record costs
    // less than a node step there: 1M triangles 73.0 -> 69.9 ms per pass, image 369 -> 271 MB; profiles/r03_octree_params.txt)
    out.max_leaf = (accel && accel->max_leaf_prims) ? accel->max_leaf_prims : (nprim > 1024u ? 8u : 4u);
    out.max_depth = (accel && accel->max_depth) ? std::min(accel->max_depth, MAX_TREE_DEPTH) : 10u;
    out.stay = nprim > 1024u ? 4.0f : 1.0f;
    out.pad_rel = (accel && accel->pad_rel > 0.0f) ? accel->pad_rel : 1.0e-4f;
    uint32_t nnode = 0;
    float pad = 0.0f;
    float root_cell_v[6] = { 0, 0, 0, 0, 0, 0 };
    out.node_box.clear(); out.node_words.clear(); out.refs.clear();
    std::vector<float> node_cell;     // per flat node: cell min xyz, 2^-depth
    if (build_tree) {
    // ---- primitive bounds ------------------------------------------------------------
    std::vector<Box> pbox(nprim);
    std::vector<TriD> ptri(nprim);
    std::vector<uint8_t> is_tri(nprim, 0);
    Box root = empty_box();
    std::vector<uint32_t> members;
    members.reserve(nprim);
    for (uint32_t i = 0; i < nobj; ++i) {
        const mcrt_object& o = sc->objects[i];
        if (o.kind == MCRT_OBJECT_SPHERE) {
            Box& b = pbox[first_prim[i]];
            float r = std::fabs(o.radius);
            for (int a = 0; a < 3; ++a) { b.mn[a] = o.position[a] - r; b.mx[a] = o.position[a] + r; }
            // radius < FLT_EPSILON never intersects (source/objects/sphere.h:18): keep it out of the tree
            if (o.radius >= FLT_EPSILON) { grow(root, b); members.push_back(first_prim[i]); }
        } else {
            for (uint32_t t = 0; t < o.triangle_count; ++t) {
                const mcrt_triangle& tr = sc->triangles[o.first_triangle + t];
                Box& b = pbox[first_prim[i] + t];
                is_tri[first_prim[i] + t] = 1;
                for (int a = 0; a < 3; ++a) { ptri[first_prim[i] + t].v[0][a] = tr.v0[a]; ptri[first_prim[i] + t].v[1][a] = tr.v1[a]; ptri[first_prim[i] + t].v[2][a] = tr.v2[a]; }
                for (int a = 0; a < 3; ++a) {
                    b.mn[a] = std::min(tr.v0[a], std::min(tr.v1[a], tr.v2[a]));
                    b.mx[a] = std::max(tr.v0[a], std::max(tr.v1[a], tr.v2[a]));
                }
                grow(root, b);
                members.push_back(first_prim[i] + t);
            }
        }
    }
    if (members.empty()) { for (int a = 0; a < 3; ++a) { root.mn[a] = -1.0f; root.mx[a] = 1.0f; } }

    float extent = std::max(root.mx[0] - root.mn[0], std::max(root.mx[1] - root.mn[1], root.mx[2] - root.mn[2]));
    float mag = 0.0f;
    for (int a = 0; a < 3; ++a) mag = std::max(mag, std::max(std::fabs(root.mn[a]), std::fabs(root.mx[a])));
    float pad_rel = (accel && accel->pad_rel > 0.0f) ? accel->pad_rel : 1.0e-4f;
    // absolute padding: a fraction of the scene extent, never below a few hundred ulps of the largest coordinate
    pad = std::max(pad_rel * std::max(extent, 1.0e-3f), 256.0f * FLT_EPSILON * mag);
    for (Box& b : pbox) for (int a = 0; a < 3; ++a) { b.mn[a] -= pad; b.mx[a] += pad; }
    for (int a = 0; a < 3; ++a) { root.mn[a] -= 2.0f * pad; root.mx[a] += 2.0f * pad; }

    // ---- build -------------------------------------------------------------------------
    Builder B{ pbox, ptri, is_tri, pad, nprim > 1024u ? 4.0f : 1.0f };
    B.max_leaf = out.max_leaf;
    B.max_depth = (accel && accel->max_depth) ? std::min(accel->max_depth, MAX_TREE_DEPTH) : 10u;
    B.pool.reserve(1024);
    B.make(root, members, 0);
    if (B.pool.size() >= (1u << 24)) { err = "octree has more than 2^24 nodes"; return MCRT_ERR_LIMIT; }

    // ---- flatten breadth-first: children of a node are contiguous ----------------------
    nnode = uint32_t(B.pool.size());
    std::vector<int> order;           // flat index -> build index
    std::vector<uint32_t> flat_of(nnode, 0);
    order.reserve(nnode);
    order.push_back(0);
    for (size_t head = 0; head < order.size(); ++head) {
        const BuildNode& n = B.pool[size_t(order[head])];
        for (int c = 0; c < 8; ++c)
            if (n.child[c] >= 0) { flat_of[size_t(n.child[c])] = uint32_t(order.size()); order.push_back(n.child[c]); }
    }

    out.node_box.assign(size_t(nnode) * 6, 0.0f);
    out.node_words.assign(size_t(nnode) * 2, 0u);
    out.refs.clear();
    for (uint32_t f = 0; f < nnode; ++f) {
        const BuildNode& n = B.pool[size_t(order[f])];
        Box t = n.tight;
        if (t.mn[0] > t.mx[0]) t = n.cell;   // empty subtree (only the root of an empty scene)
        for (int a = 0; a < 3; ++a) { out.node_box[size_t(f) * 6 + a] = t.mn[a]; out.node_box[size_t(f) * 6 + 3 + a] = t.mx[a]; }
        uint32_t mask = 0, base = 0;
        bool have = false;
        for (int c = 0; c < 8; ++c)
            if (n.child[c] >= 0) { mask |= 1u << c; if (!have) { base = flat_of[size_t(n.child[c])]; have = true; } }
        out.node_words[size_t(f) * 2] = (mask << 24) | base;
        if (n.own.empty()) out.node_words[size_t(f) * 2 + 1] = NO_REFS;
        else {
            out.node_words[size_t(f) * 2 + 1] = uint32_t(out.refs.size());
            for (size_t k = 0; k < n.own.size(); ++k) out.refs.push_back(n.own[k] | (k + 1 == n.own.size() ? REF_LAST : 0u));
        }
    }
    out.tree_depth = B.deepest;
    for (int a = 0; a < 3; ++a) { root_cell_v[a] = B.pool[0].cell.mn[a]; root_cell_v[3 + a] = B.pool[0].cell.mx[a]; }
    node_cell.resize(size_t(nnode) * 4);
    for (uint32_t f = 0; f < nnode; ++f) {
        const BuildNode& bn = B.pool[size_t(order[f])];
        for (int k = 0; k < 3; ++k) node_cell[size_t(f) * 4 + k] = bn.cell.mn[k];
        node_cell[size_t(f) * 4 + 3] = std::ldexp(1.0f, -int(bn.depth));
    }
    } else out.tree_depth = 0;
    out.ref_count = uint32_t(out.refs.size());
    for (int a = 0; a < 6; ++a) out.root_cell[a] = root_cell_v[a];
    out.pad = pad;

    // ---- SMALL image -----------------------------------------------------------------
    SceneView& v = out.view;
    v = SceneView{};
    v.nobj = nobj; v.nlights = uint32_t(lights.size()); v.ntri = sc->triangle_count; v.nprim = nprim; v.nnode = nnode;
    for (int a = 0; a < 3; ++a) v.root_size[a] = root_cell_v[3 + a] - root_cell_v[a];
    v.walk_pad = kWalkPadFraction * pad;
    v.far_limit = pad * kFarOriginFactor;
    v.off_obj = 0;
    v.off_lights = v.off_obj + nobj * OBJ_STRIDE;
    v.off_bg = (v.off_lights + v.nlights + 1u) & ~1u;   // doubles: even word offset
    v.small_words = v.off_bg + 6;
    out.small_img.assign(v.small_words, 0u);
    for (uint32_t i = 0; i < nobj; ++i) {
        const mcrt_object& o = sc->objects[i];
        uint32_t* w = out.small_img.data() + v.off_obj + size_t(i) * OBJ_STRIDE;
        w[OBJ_KIND] = uint32_t(o.kind);
        w[OBJ_FIRST_TRI] = o.first_triangle;
        w[OBJ_TRI_COUNT] = o.kind == MCRT_OBJECT_SPHERE ? 0u : o.triangle_count;
        w[OBJ_FIRST_PRIM] = first_prim[i];
        w[OBJ_RADIUS] = fbits(o.radius);
        for (int a = 0; a < 3; ++a) {
            w[OBJ_POS + a] = fbits(o.position[a]);
            w[OBJ_AABB_MIN + a] = fbits(o.aabb_min[a]);
            w[OBJ_AABB_MAX + a] = fbits(o.aabb_max[a]);
            w[OBJ_LIGHT_X + a] = fbits(o.light_x[a]);
            w[OBJ_LIGHT_Y + a] = fbits(o.light_y[a]);
        }
        w[OBJ_SURFACE] = uint32_t(o.material.surface);
        w[OBJ_DIFFUSE] = uint32_t(o.material.diffuse);
        w[OBJ_ALBEDO] = fbits(o.material.albedo);
        w[OBJ_ROUGHNESS] = fbits(o.material.roughness);
        w[OBJ_IOR] = fbits(o.material.refractive_index);
        {
            float A, B;
            oren_nayar_terms(o.material.roughness, A, B);
            w[OBJ_ON_A] = fbits(A);
            w[OBJ_ON_B] = fbits(B);
        }
        for (int a = 0; a < 3; ++a) {
            put_double(out.small_img, v.off_obj + size_t(i) * OBJ_STRIDE + OBJ_COLOR + 2 * a, o.material.color[a]);
            put_double(out.small_img, v.off_obj + size_t(i) * OBJ_STRIDE + OBJ_EMISSION + 2 * a, o.material.emission[a]);
        }
    }
    for (uint32_t k = 0; k < v.nlights; ++k) out.small_img[v.off_lights + k] = lights[k];
    for (int a = 0; a < 3; ++a) put_double(out.small_img, v.off_bg + 2 * a, sc->background[a]);

    // ---- BIG image -------------------------------------------------------------------
    uint64_t off = 0;
    v.off_node = uint32_t(off);      off += uint64_t(nnode) * NODE_STRIDE;
    v.off_leaf = uint32_t(off);      if (build_tree) off += (uint64_t(out.refs.size()) + 1u) * LEAF_STRIDE;   // +1: the traversal prefetches one record ahead
    v.off_tri = uint32_t(off);       off += uint64_t(sc->triangle_count) * TRI_STRIDE;
    v.tri_src = v.off_tri;
    // triangle-pair records for the scan path: only built for scenes it can be chosen for
    std::vector<uint32_t> first_pair(nobj, 0);
    uint64_t npair = 0;
    const bool want_pairs = sc->triangle_count <= MCRT_SCAN_PAIR_LIMIT;
    v.has_pairs = want_pairs ? 1u : 0u;
    if (want_pairs)
        for (uint32_t i = 0; i < nobj; ++i) {
            first_pair[i] = uint32_t(npair);
            if (sc->objects[i].kind != MCRT_OBJECT_SPHERE) npair += (sc->objects[i].triangle_count + 1u) / 2u;
        }
    v.off_pair = uint32_t(off);      off += npair * PAIR_STRIDE;
    if (off >= 0xFFFFFFF0ull) { err = "scene image exceeds 16 GiB of words"; return MCRT_ERR_LIMIT; }
    v.big_words = uint32_t(off);
    out.big_img.assign(v.big_words, 0u);
    for (uint32_t f = 0; f < nnode; ++f) {
        uint32_t* w = out.big_img.data() + v.off_node + size_t(f) * NODE_STRIDE;
        for (uint32_t k = 0; k < 6; ++k) w[NODE_BOX + k] = fbits(out.node_box[size_t(f) * 6 + k]);
        w[NODE_CHILD] = out.node_words[size_t(f) * 2];
        w[NODE_LEAF] = out.node_words[size_t(f) * 2 + 1];
        for (uint32_t k = 0; k < 4; ++k) w[NODE_CELL + k] = fbits(node_cell[size_t(f) * 4 + k]);
        if (w[NODE_LEAF] != NO_REFS) {          // the length of the node's own list rides in the mantissa of the scale word
            uint32_t n = 1;
            for (size_t r = w[NODE_LEAF]; !(out.refs[r] & REF_LAST); ++r) ++n;
            w[NODE_SCALE] |= n & NODE_COUNT_MASK;
        }
    }
    for (uint32_t t = 0; t < sc->triangle_count; ++t) {
        const mcrt_triangle& tr = sc->triangles[t];
        uint32_t* w = out.big_img.data() + v.off_tri + size_t(t) * TRI_STRIDE;
        for (int a = 0; a < 3; ++a) {
            // the two subtractions of source/core/triangle.h:31-32, done once (plain IEEE float subtraction)
            volatile float e1 = tr.v1[a] - tr.v0[a];
            volatile float e2 = tr.v2[a] - tr.v0[a];
            w[TRI_V0 + a] = fbits(tr.v0[a]);
            w[TRI_E1 + a] = fbits(e1);
            w[TRI_E2 + a] = fbits(e2);
            w[TRI_N + a] = fbits(tr.normal[a]);
        }
    }
    for (size_t r = 0; r < out.refs.size(); ++r) {
        const uint32_t prim = out.refs[r] & ~REF_LAST;
        const uint32_t obj = prim_obj[prim];
        const mcrt_object& o = sc->objects[obj];
        uint32_t* w = out.big_img.data() + v.off_leaf + r * LEAF_STRIDE;
        w[LEAF_PRIM] = out.refs[r];
        if (o.kind == MCRT_OBJECT_SPHERE) w[LEAF_OBJ] = obj | LEAF_SPHERE;
        else {
            w[LEAF_OBJ] = obj;
            const uint32_t* tr = out.big_img.data() + v.off_tri + size_t(o.first_triangle + (prim - first_prim[obj])) * TRI_STRIDE;
            for (uint32_t k = 0; k < 9; ++k) w[k] = tr[k];   // v0, e1, e2 are words 0..8 of both records
        }
    }
    if (want_pairs)
        for (uint32_t i = 0; i < nobj; ++i) {
            const mcrt_object& o = sc->objects[i];
            out.small_img[v.off_obj + size_t(i) * OBJ_STRIDE + OBJ_FIRST_PAIR] = first_pair[i];
            if (o.kind == MCRT_OBJECT_SPHERE) continue;
            for (uint32_t j = 0; j * 2 < o.triangle_count; ++j) {
                const uint32_t ta = o.first_triangle + 2 * j;
                const uint32_t tb = (2 * j + 1 < o.triangle_count) ? ta + 1 : ta;
                const uint32_t* ra = out.big_img.data() + v.off_tri + size_t(ta) * TRI_STRIDE;
                const uint32_t* rb = out.big_img.data() + v.off_tri + size_t(tb) * TRI_STRIDE;
                uint32_t* w = out.big_img.data() + v.off_pair + size_t(first_pair[i] + j) * PAIR_STRIDE;
                for (uint32_t k = 0; k < 9; ++k) { w[2 * k] = ra[k]; w[2 * k + 1] = rb[k]; }   // v0, e1, e2 are words 0..8 of a record
            }
        }
    return MCRT_OK;
}

} // namespace mcrt
