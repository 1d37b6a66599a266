// mcrt_build.h - the flat octree built ON THE DEVICE.
//
// Replaces, for scenes edited between frames, the host build of csrc/mcrt_flatten.cpp (itself the replacement of
// Octree::Fill / Subdivide / InsertIfOverlaps, source/accelerationstructures/octree.h:50-92,168-223): the same tree -
// same cells, same split rule, same overlap tests, same breadth-first node numbering - produced level by level by
// kernels, so that a 1M-triangle scene is re-indexed in tens of milliseconds instead of 1.7 s of host time.
//
// One level = every node created by the level before, with its members (primitive ids) contiguous in a level array:
//   k_bld_classify   one thread per (node, member): the member's box clipped to the node's padded cell goes into the
//                    node's tight box (atomic min / max on order-preserving integers); a terminal node (<= max_leaf
//                    members, or max_depth reached) keeps all its members; otherwise a member wider than `stay` child
//                    cells stays in the node's own list and the rest are tested against the eight child cells
//                    (padded-box overlap, then the triangle/box separating-axis test in double, as the host does);
//   k_bld_decide     one thread per node: "a split that repeats every member in every child separates nothing"
//                    (octree.h:195-212) - then the node keeps everything; else its non-empty octants become children;
//   exclusive scans  (hipCUB) give every child its node index (children of a node contiguous, octant order, levels one
//                    after the other: the breadth-first numbering of the host flattener), every child its slice of the
//                    next level array and every node its slice of the leaf-record array;
//   k_bld_children   one thread per (node, octant): the child's cell from its parent's (the host's float formulas);
//   k_bld_scatter    one thread per (node, member): member ids into the children's slices and into the node's own list.
// k_bld_emit_nodes / k_bld_emit_leaves then write the node and leaf records of csrc/mcrt_layout.h.
// The order of the entries inside a list depends on atomics; nothing that is computed from the tree depends on it (the
// nearest hit is the minimum of (t, primitive id) over the candidates).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>
#include <vector>

#include "mcrt_layout.h"

namespace mcrt {
namespace bld {

constexpr int kThreads = 256;
constexpr uint32_t OWN_BIT = 0x100u;

struct Params {
    uint32_t nobj, nprim, ntri;
    uint32_t max_leaf, max_depth;
    float stay, pad_rel;
};

// everything the kernels share; arrays are device pointers
struct Nodes {
    float* cell;          // [cap][6]: cell min xyz, max xyz
    int* tight;           // [cap][6]: order-preserving integers of the tight box (min xyz, max xyz)
    uint32_t* depth;      // [cap]
    uint32_t* begin;      // [cap]: first member in the node's level array
    uint32_t* count;      // [cap]: members
    uint32_t* own_begin;  // [cap]: first leaf record
    uint32_t* own_count;  // [cap]
    uint32_t* child_word; // [cap]: mask << 24 | first child
};

__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i ^ ((i >> 31) & 0x7FFFFFFF); }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7FFFFFFF)); }

// root[0..5]: order-preserving integers of the union of the primitive boxes; root[6]: order-preserving integer of the largest |coordinate|
__global__ void k_bld_boxes(Params P, const uint32_t* __restrict__ small_img, uint32_t off_obj, const float* __restrict__ raw_tris, float* __restrict__ pbox,
                            uint32_t* __restrict__ prim_obj, uint8_t* __restrict__ prim_skip, int* __restrict__ root)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.nprim) return;
    // the object that owns primitive p (ids are assigned in (object, element) order; a few hundred objects at most)
    uint32_t obj = 0;
    for (uint32_t i = 0; i < P.nobj; ++i) if (small_img[off_obj + i * OBJ_STRIDE + OBJ_FIRST_PRIM] <= p) obj = i;
    const uint32_t* o = small_img + off_obj + obj * OBJ_STRIDE;
    prim_obj[p] = obj;
    float mn[3], mx[3];
    bool skip = false;
    if (o[OBJ_KIND] == (uint32_t)MCRT_OBJECT_SPHERE) {
        const float radius = __uint_as_float(o[OBJ_RADIUS]), r = fabsf(radius);
        for (int a = 0; a < 3; ++a) { const float c = __uint_as_float(o[OBJ_POS + a]); mn[a] = c - r; mx[a] = c + r; }
        skip = !(radius >= 1.1920928955078125e-07f);          // never intersects (source/objects/sphere.h:18): kept out of the tree
    } else {
        const float* t = raw_tris + (size_t)(o[OBJ_FIRST_TRI] + (p - o[OBJ_FIRST_PRIM])) * 12;
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(t[a], fminf(t[3 + a], t[6 + a]));
            mx[a] = fmaxf(t[a], fmaxf(t[3 + a], t[6 + a]));
        }
    }
    for (int a = 0; a < 3; ++a) { pbox[(size_t)p * 6 + a] = mn[a]; pbox[(size_t)p * 6 + 3 + a] = mx[a]; }
    prim_skip[p] = skip ? 1 : 0;
    if (!skip)
        for (int a = 0; a < 3; ++a) { atomicMin(root + a, f2ord(mn[a])); atomicMax(root + 3 + a, f2ord(mx[a])); }
}

// pad, root cell, node 0 (the host's formulas, csrc/mcrt_flatten.cpp); out[0..5] root cell, out[6] pad
__global__ void k_bld_root(Params P, const int* __restrict__ root, uint32_t members, Nodes N, float* __restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(root[a]); mx[a] = ord2f(root[3 + a]); }
    if (members == 0) for (int a = 0; a < 3; ++a) { mn[a] = -1.0f; mx[a] = 1.0f; }
    const float extent = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
    float mag = 0.0f;
    for (int a = 0; a < 3; ++a) mag = fmaxf(mag, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
    const float pad = fmaxf(P.pad_rel * fmaxf(extent, 1.0e-3f), 256.0f * 1.1920928955078125e-07f * mag);
    for (int a = 0; a < 3; ++a) { mn[a] -= 2.0f * pad; mx[a] += 2.0f * pad; }
    for (int a = 0; a < 3; ++a) { N.cell[a] = mn[a]; N.cell[3 + a] = mx[a]; out[a] = mn[a]; out[3 + a] = mx[a]; }
    out[6] = pad;
    for (int a = 0; a < 3; ++a) { N.tight[a] = 0x7FFFFFFF; N.tight[3 + a] = (int)0x80000000; }
    N.depth[0] = 0; N.begin[0] = 0; N.count[0] = members;
}

// the members of the root: every primitive that can be hit (order-preserving compaction is not needed)
__global__ void k_bld_root_members(uint32_t nprim, const uint8_t* __restrict__ prim_skip, uint32_t* __restrict__ mem_prim, uint32_t* __restrict__ mem_node,
                                   uint32_t* __restrict__ counter)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool keep = p < nprim && !prim_skip[p];
    const uint64_t m = __ballot(keep);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(counter, (uint32_t)__popcll(m));
    at = (uint32_t)__shfl((int)at, (int)leader);
    if (keep) { const uint32_t i = at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); mem_prim[i] = p; mem_node[i] = 0u; }
}

// Triangle / box overlap by separating axes, in double: tri_overlaps_box of csrc/mcrt_flatten.cpp, operation by operation.
__device__ inline bool tri_overlaps_box(const float* t, const float* bmn, const float* bmx)
{
    double c[3], h[3], v[3][3];
    for (int a = 0; a < 3; ++a) { c[a] = 0.5 * ((double)bmn[a] + (double)bmx[a]); h[a] = 0.5 * ((double)bmx[a] - (double)bmn[a]); }
    for (int i = 0; i < 3; ++i) for (int a = 0; a < 3; ++a) v[i][a] = (double)t[i * 3 + a] - c[a];
    const double e[3][3] = { { v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2] },
                             { v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2] },
                             { v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2] } };
    const double eps = 1e-12;
    for (int i = 0; i < 3; ++i)
        for (int a = 0; a < 3; ++a) {
            const int b = (a + 1) % 3, cc = (a + 2) % 3;
            const double ax_b = -e[i][cc], ax_c = e[i][b];
            double lo = 1e300, hi = -1e300;
            for (int k = 0; k < 3; ++k) { const double p = ax_b * v[k][b] + ax_c * v[k][cc]; lo = fmin(lo, p); hi = fmax(hi, p); }
            const double r = h[b] * fabs(ax_b) + h[cc] * fabs(ax_c) + eps;
            if (lo > r || hi < -r) return false;
        }
    for (int a = 0; a < 3; ++a) {
        const double lo = fmin(v[0][a], fmin(v[1][a], v[2][a])), hi = fmax(v[0][a], fmax(v[1][a], v[2][a]));
        if (lo > h[a] + eps || hi < -h[a] - eps) return false;
    }
    const double n[3] = { e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0] };
    const double d = n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2];
    const double r = h[0] * fabs(n[0]) + h[1] * fabs(n[1]) + h[2] * fabs(n[2]) + eps;
    return fabs(d) <= r;
}

// per level: cnt[(node - first) * 8 + c] members that go to octant c, own[(node - first)] members that stay, placed / down as in Builder::make
__global__ void k_bld_classify(Params P, float pad, uint32_t first_node, uint32_t members, const uint32_t* __restrict__ mem_prim, const uint32_t* __restrict__ mem_node,
                               uint16_t* __restrict__ mem_mask, Nodes N, const float* __restrict__ pbox, const uint8_t* __restrict__ is_sphere,
                               const float* __restrict__ raw_tris, const uint32_t* __restrict__ small_img, uint32_t off_obj, const uint32_t* __restrict__ prim_obj,
                               uint32_t* __restrict__ cnt, uint32_t* __restrict__ own, uint32_t* __restrict__ placed, uint32_t* __restrict__ down)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= members) return;
    const uint32_t node = mem_node[m], p = mem_prim[m], ln = node - first_node;
    float bmn[3], bmx[3], cmn[3], cmx[3];
    for (int a = 0; a < 3; ++a) {
        bmn[a] = pbox[(size_t)p * 6 + a] - pad; bmx[a] = pbox[(size_t)p * 6 + 3 + a] + pad;       // the padded primitive box
        cmn[a] = N.cell[(size_t)node * 6 + a]; cmx[a] = N.cell[(size_t)node * 6 + 3 + a];
    }
    // the node's tight box: union of (padded member box) clipped to the PADDED cell.  Members of a node are contiguous, so on
    // the upper levels a whole wavefront belongs to one node: then one atomic per wavefront instead of 64 on the same address.
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t active = __ballot(1);
    const uint32_t node0 = (uint32_t)__shfl((int)node, (int)(__ffsll((unsigned long long)active) - 1));
    const bool uniform = __ballot(node == node0) == active && active == ~0ull;
    for (int a = 0; a < 3; ++a) {
        int lo = f2ord(fmaxf(bmn[a], cmn[a] - pad)), hi = f2ord(fminf(bmx[a], cmx[a] + pad));
        if (uniform) {
            for (int s = 32; s > 0; s >>= 1) { lo = min(lo, __shfl_xor(lo, s)); hi = max(hi, __shfl_xor(hi, s)); }
            if (lane == 0) { atomicMin(N.tight + (size_t)node * 6 + a, lo); atomicMax(N.tight + (size_t)node * 6 + 3 + a, hi); }
        } else {
            atomicMin(N.tight + (size_t)node * 6 + a, lo);
            atomicMax(N.tight + (size_t)node * 6 + 3 + a, hi);
        }
    }
    const uint32_t depth = N.depth[node];
    uint32_t mask = OWN_BIT;
    if (!(N.count[node] <= P.max_leaf || depth >= P.max_depth)) {
        float half[3], centre[3];
        for (int a = 0; a < 3; ++a) { half[a] = 0.5f * (cmx[a] - cmn[a]); centre[a] = cmn[a] + half[a]; }
        const float child_extent = fmaxf(half[0], fmaxf(half[1], half[2]));
        const float ext = fmaxf(bmx[0] - bmn[0], fmaxf(bmx[1] - bmn[1], bmx[2] - bmn[2]));
        if (!(ext > P.stay * child_extent)) {
            mask = 0;
            const uint32_t obj = prim_obj[p];
            const uint32_t* o = small_img + off_obj + obj * OBJ_STRIDE;
            const bool tri = o[OBJ_KIND] != (uint32_t)MCRT_OBJECT_SPHERE;
            const float* t = tri ? raw_tris + (size_t)(o[OBJ_FIRST_TRI] + (p - o[OBJ_FIRST_PRIM])) * 12 : nullptr;
            for (uint32_t c = 0; c < 8; ++c) {
                float kmn[3], kmx[3];
                bool over = true;
                for (int a = 0; a < 3; ++a) {
                    const bool hi = (c >> a) & 1u;
                    kmn[a] = hi ? centre[a] : cmn[a];
                    kmx[a] = hi ? cmx[a] : centre[a];
                    if (bmx[a] < kmn[a] || kmx[a] < bmn[a]) over = false;
                }
                if (!over) continue;
                if (tri) {
                    float pmn[3], pmx[3];
                    for (int a = 0; a < 3; ++a) { pmn[a] = kmn[a] - pad; pmx[a] = kmx[a] + pad; }
                    if (!tri_overlaps_box(t, pmn, pmx)) continue;
                }
                mask |= 1u << c;
            }
        }
    }
    // counts: cnt per octant, own, placed, down (Builder::make's `placed` and `down.size()`)
    const bool is_down = !(mask & OWN_BIT);
    if (uniform) {
        for (uint32_t c = 0; c < 8; ++c) {
            const uint64_t b = __ballot(is_down && (mask & (1u << c)));
            if (b && lane == 0) atomicAdd(cnt + (size_t)ln * 8 + c, (uint32_t)__popcll(b));
        }
        uint32_t pl = is_down ? (uint32_t)__popc(mask) : 0u;
        for (int s = 32; s > 0; s >>= 1) pl += (uint32_t)__shfl_xor((int)pl, s);
        const uint64_t bd = __ballot(is_down), bo = __ballot(!is_down);
        if (lane == 0) {
            if (pl) atomicAdd(placed + ln, pl);
            if (bd) atomicAdd(down + ln, (uint32_t)__popcll(bd));
            if (bo) atomicAdd(own + ln, (uint32_t)__popcll(bo));
        }
    } else {
        if (is_down) {
            for (uint32_t c = 0; c < 8; ++c) if (mask & (1u << c)) atomicAdd(cnt + (size_t)ln * 8 + c, 1u);
            atomicAdd(placed + ln, (uint32_t)__popc(mask));
            atomicAdd(down + ln, 1u);
        } else atomicAdd(own + ln, 1u);
    }
    mem_mask[m] = (uint16_t)mask;
    (void)is_sphere;
}

// per node of the level: does the split stand?  kids[ln] = number of children, own[ln] = final own count, cnt zeroed where the split fails
__global__ void k_bld_decide(uint32_t first_node, uint32_t nodes, Nodes N, uint32_t* __restrict__ cnt, uint32_t* __restrict__ own, const uint32_t* __restrict__ placed,
                             const uint32_t* __restrict__ down, uint32_t* __restrict__ kids, uint8_t* __restrict__ failed)
{
    const uint32_t ln = blockIdx.x * blockDim.x + threadIdx.x;
    if (ln >= nodes) return;
    const uint32_t node = first_node + ln;
    const bool fail = down[ln] > 0u && (uint64_t)placed[ln] >= (uint64_t)down[ln] * 8ull;
    uint32_t mask = 0;
    if (fail) {
        for (uint32_t c = 0; c < 8; ++c) cnt[(size_t)ln * 8 + c] = 0u;
        own[ln] = N.count[node];
    } else {
        for (uint32_t c = 0; c < 8; ++c) if (cnt[(size_t)ln * 8 + c] > 0u) mask |= 1u << c;
    }
    failed[ln] = fail ? 1 : 0;
    kids[ln] = (uint32_t)__popc(mask);
    N.own_count[node] = own[ln];
    N.child_word[node] = mask << 24;        // the first-child index is added by k_bld_children
}

// per (node, octant) with members: the child node.  kid_off / mem_off / own_off: exclusive scans of kids / cnt / own over the level.
__global__ void k_bld_children(uint32_t first_node, uint32_t nodes, uint32_t next_first, uint32_t leaf_base, Nodes N, const uint32_t* __restrict__ cnt,
                               const uint32_t* __restrict__ kid_off, const uint32_t* __restrict__ mem_off, const uint32_t* __restrict__ own_off)
{
    const uint32_t ln = blockIdx.x * blockDim.x + threadIdx.x;
    if (ln >= nodes) return;
    const uint32_t node = first_node + ln;
    N.own_begin[node] = leaf_base + own_off[ln];
    const uint32_t mask = N.child_word[node] >> 24;
    if (!mask) return;
    const uint32_t base = next_first + kid_off[ln];
    N.child_word[node] = (mask << 24) | base;
    float cmn[3], cmx[3], centre[3];
    for (int a = 0; a < 3; ++a) {
        cmn[a] = N.cell[(size_t)node * 6 + a]; cmx[a] = N.cell[(size_t)node * 6 + 3 + a];
        const float half = 0.5f * (cmx[a] - cmn[a]);
        centre[a] = cmn[a] + half;
    }
    uint32_t rank = 0;
    for (uint32_t c = 0; c < 8; ++c) {
        if (!(mask & (1u << c))) continue;
        const uint32_t ch = base + rank++;
        for (int a = 0; a < 3; ++a) {
            const bool hi = (c >> a) & 1u;
            N.cell[(size_t)ch * 6 + a] = hi ? centre[a] : cmn[a];
            N.cell[(size_t)ch * 6 + 3 + a] = hi ? cmx[a] : centre[a];
            N.tight[(size_t)ch * 6 + a] = 0x7FFFFFFF; N.tight[(size_t)ch * 6 + 3 + a] = (int)0x80000000;
        }
        N.depth[ch] = N.depth[node] + 1u;
        N.begin[ch] = mem_off[(size_t)ln * 8 + c];
        N.count[ch] = cnt[(size_t)ln * 8 + c];
    }
}

// per (node, member): into the children's slices of the next level array and into the node's own list (refs: primitive id, REF_LAST on the last entry)
__global__ void k_bld_scatter(uint32_t first_node, uint32_t members, const uint32_t* __restrict__ mem_prim, const uint32_t* __restrict__ mem_node,
                              const uint16_t* __restrict__ mem_mask, Nodes N, const uint8_t* __restrict__ failed, uint32_t* __restrict__ fill, uint32_t* __restrict__ fill_own,
                              uint32_t* __restrict__ next_prim, uint32_t* __restrict__ next_node, uint32_t* __restrict__ refs, uint32_t leaf_base)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= members) return;
    const uint32_t node = mem_node[m], p = mem_prim[m], ln = node - first_node;
    uint32_t mask = mem_mask[m];
    if (failed[ln]) mask = OWN_BIT;
    if (mask & OWN_BIT) {
        const uint32_t pos = atomicAdd(fill_own + ln, 1u), n = N.own_count[node];
        refs[N.own_begin[node] - leaf_base + pos] = p | (pos + 1u == n ? REF_LAST : 0u);     // refs: this level's part of the list array
        return;
    }
    const uint32_t cw = N.child_word[node], cmask = cw >> 24, base = cw & 0x00FFFFFFu;
    for (uint32_t c = 0; c < 8; ++c) {
        if (!(mask & (1u << c))) continue;
        const uint32_t ch = base + (uint32_t)__popc(cmask & ((1u << c) - 1u));
        const uint32_t pos = atomicAdd(fill + (size_t)ln * 8 + c, 1u);
        next_prim[N.begin[ch] + pos] = p;
        next_node[N.begin[ch] + pos] = ch;
    }
}

__global__ void k_bld_emit_nodes(uint32_t nodes, Nodes N, uint32_t* __restrict__ big, uint32_t off_node)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nodes) return;
    uint32_t* w = big + off_node + (size_t)f * NODE_STRIDE;
    float t[6];
    for (int k = 0; k < 6; ++k) t[k] = ord2f(N.tight[(size_t)f * 6 + k]);
    if (t[0] > t[3]) for (int k = 0; k < 6; ++k) t[k] = N.cell[(size_t)f * 6 + k];      // empty subtree (only the root of an empty scene)
    for (int k = 0; k < 6; ++k) w[NODE_BOX + k] = __float_as_uint(t[k]);
    w[NODE_CHILD] = N.child_word[f];
    w[NODE_LEAF] = N.own_count[f] ? N.own_begin[f] : NO_REFS;
    for (int k = 0; k < 3; ++k) w[NODE_CELL + k] = __float_as_uint(N.cell[(size_t)f * 6 + k]);
    w[NODE_SCALE] = __float_as_uint(ldexpf(1.0f, -(int)N.depth[f])) | (N.own_count[f] & NODE_COUNT_MASK);
}

__global__ void k_bld_emit_leaves(uint32_t nrefs, const uint32_t* __restrict__ refs, const uint32_t* __restrict__ prim_obj, const uint32_t* __restrict__ small_img,
                                  uint32_t off_obj, uint32_t* __restrict__ big, uint32_t off_leaf, uint32_t off_tri)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrefs) return;
    uint32_t* w = big + off_leaf + (size_t)r * LEAF_STRIDE;
    if (r == nrefs) { for (uint32_t k = 0; k < LEAF_STRIDE; ++k) w[k] = 0u; return; }     // the pad record the walk prefetches
    const uint32_t e = refs[r], prim = e & ~REF_LAST, obj = prim_obj[prim];
    const uint32_t* o = small_img + off_obj + obj * OBJ_STRIDE;
    for (uint32_t k = 0; k < LEAF_STRIDE; ++k) w[k] = 0u;
    w[LEAF_PRIM] = e;
    if (o[OBJ_KIND] == (uint32_t)MCRT_OBJECT_SPHERE) w[LEAF_OBJ] = obj | LEAF_SPHERE;
    else {
        w[LEAF_OBJ] = obj;
        const uint32_t* tr = big + off_tri + (size_t)(o[OBJ_FIRST_TRI] + (prim - o[OBJ_FIRST_PRIM])) * TRI_STRIDE;
        for (uint32_t k = 0; k < 9; ++k) w[k] = tr[k];
    }
}

// read the tree back out of the records (mcrt_octree_download after a device build)
__global__ void k_bld_unpack(uint32_t nodes, uint32_t nrefs, const uint32_t* __restrict__ big, uint32_t off_node, uint32_t off_leaf, float* __restrict__ node_box,
                             uint32_t* __restrict__ node_words, uint32_t* __restrict__ refs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nodes) {
        const uint32_t* w = big + off_node + (size_t)i * NODE_STRIDE;
        for (int k = 0; k < 6; ++k) node_box[(size_t)i * 6 + k] = __uint_as_float(w[NODE_BOX + k]);
        node_words[(size_t)i * 2] = w[NODE_CHILD]; node_words[(size_t)i * 2 + 1] = w[NODE_LEAF];
    }
    if (i < nrefs) refs[i] = big[off_leaf + (size_t)i * LEAF_STRIDE + LEAF_PRIM];
}

template <class T> struct Buf {
    T* p = nullptr;
    size_t n = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { n = count; return hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)); }
};

#define BLD_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; }  \
    } while (0)

struct Result {
    uint32_t* d_big = nullptr;      // the complete BIG image (caller owns it)
    uint32_t nnode = 0, nrefs = 0, depth = 0;
    uint32_t off_node = 0, off_leaf = 0, off_tri = 0, off_pair = 0;
    uint64_t big_words = 0;
    float root_cell[6] = { 0, 0, 0, 0, 0, 0 };
    float pad = 0.0f;
    float build_ms = 0.0f;
};

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)std::max<uint64_t>(1, (n + kThreads - 1) / kThreads); }

// d_small: the SMALL image on the device; raw_tris: the caller's mcrt_triangle array (host); tail: the triangle + pair records the host
// flattener laid out from word 0 (tail_off_pair = word offset of the pair records inside it).  Returns 0, or 1 with `err` set.
inline int build_on_device(hipStream_t stream, const Params& P, const uint32_t* d_small, uint32_t off_obj, const float* h_raw_tris, const std::vector<uint32_t>& tail,
                           uint32_t tail_off_pair, Result& R, std::string& err)
{
    struct Events {                     // destroyed on every way out (BLD_TRY returns early)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } events;
    BLD_TRY(hipEventCreate(&events.a)); BLD_TRY(hipEventCreate(&events.b));
    const hipEvent_t ev0 = events.a, ev1 = events.b;
    Buf<float> raw, pbox, rootinfo;
    Buf<uint32_t> prim_obj, counter;
    Buf<uint8_t> skip;
    Buf<int> rootbox;
    BLD_TRY(raw.alloc((size_t)P.ntri * 12)); BLD_TRY(pbox.alloc((size_t)P.nprim * 6)); BLD_TRY(rootinfo.alloc(8));
    BLD_TRY(prim_obj.alloc(P.nprim)); BLD_TRY(skip.alloc(P.nprim)); BLD_TRY(rootbox.alloc(8)); BLD_TRY(counter.alloc(4));
    if (P.ntri) BLD_TRY(hipMemcpyAsync(raw.p, h_raw_tris, (size_t)P.ntri * 48, hipMemcpyHostToDevice, stream));
    BLD_TRY(hipEventRecord(ev0, stream));
    const int init_box[8] = { 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0 };
    BLD_TRY(hipMemcpyAsync(rootbox.p, init_box, sizeof init_box, hipMemcpyHostToDevice, stream));
    BLD_TRY(hipMemsetAsync(counter.p, 0, 16, stream));
    hipLaunchKernelGGL(k_bld_boxes, dim3(blocks_for(P.nprim)), dim3(kThreads), 0, stream, P, d_small, off_obj, raw.p, pbox.p, prim_obj.p, skip.p, rootbox.p);

    // node arrays: 4 nodes per primitive to begin with (the 1M-triangle soup needs 2.4), grown level by level when a scene needs more
    // (one whose padding is large against its cells - far from the origin, say - references a primitive from many of them)
    size_t cap = (size_t)P.nprim * 4 + 4096;
    Buf<float> cell; Buf<int> tight; Buf<uint32_t> depth, begin, count, own_begin, own_count, child_word;
    BLD_TRY(cell.alloc(cap * 6)); BLD_TRY(tight.alloc(cap * 6)); BLD_TRY(depth.alloc(cap)); BLD_TRY(begin.alloc(cap)); BLD_TRY(count.alloc(cap));
    BLD_TRY(own_begin.alloc(cap)); BLD_TRY(own_count.alloc(cap)); BLD_TRY(child_word.alloc(cap));
    Nodes N = { cell.p, tight.p, depth.p, begin.p, count.p, own_begin.p, own_count.p, child_word.p };

    Buf<uint32_t> mem_prim[2], mem_node[2];
    BLD_TRY(mem_prim[0].alloc(P.nprim)); BLD_TRY(mem_node[0].alloc(P.nprim));
    hipLaunchKernelGGL(k_bld_root_members, dim3(blocks_for(P.nprim)), dim3(kThreads), 0, stream, P.nprim, skip.p, mem_prim[0].p, mem_node[0].p, counter.p);
    uint32_t members = 0;
    BLD_TRY(hipMemcpyAsync(&members, counter.p, 4, hipMemcpyDeviceToHost, stream));
    BLD_TRY(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(k_bld_root, dim3(1), dim3(64), 0, stream, P, rootbox.p, members, N, rootinfo.p);
    float rootinfo_h[8];
    BLD_TRY(hipMemcpyAsync(rootinfo_h, rootinfo.p, 7 * sizeof(float), hipMemcpyDeviceToHost, stream));
    BLD_TRY(hipStreamSynchronize(stream));
    const float pad = rootinfo_h[6];

    // the leaf-record ids (refs), grown level by level on the host side of the loop
    std::vector<Buf<uint32_t>*> ref_parts;     // one array per level, concatenated at the end
    std::vector<uint32_t> ref_counts;
    struct Cleanup { std::vector<Buf<uint32_t>*>& v; ~Cleanup() { for (auto* b : v) delete b; } } cleanup{ ref_parts };

    uint32_t first_node = 0, level_nodes = 1, total_nodes = 1, total_refs = 0, level = 0, deepest = 0;
    int cur = 0;
    Buf<uint8_t> scan_tmp;
    size_t scan_tmp_bytes = 0;
    while (level_nodes > 0) {
        deepest = level;
        Buf<uint32_t> cnt, own, placed, down, kids, kid_off, mem_off, own_off, fill, fill_own, totals;
        Buf<uint16_t> mask;
        Buf<uint8_t> failed;
        const size_t ln = level_nodes;
        BLD_TRY(cnt.alloc(ln * 8 + 1)); BLD_TRY(own.alloc(ln + 1)); BLD_TRY(placed.alloc(ln)); BLD_TRY(down.alloc(ln)); BLD_TRY(kids.alloc(ln + 1));
        BLD_TRY(kid_off.alloc(ln + 1)); BLD_TRY(mem_off.alloc(ln * 8 + 1)); BLD_TRY(own_off.alloc(ln + 1)); BLD_TRY(fill.alloc(ln * 8)); BLD_TRY(fill_own.alloc(ln));
        BLD_TRY(mask.alloc(members)); BLD_TRY(failed.alloc(ln)); BLD_TRY(totals.alloc(4));
        BLD_TRY(hipMemsetAsync(cnt.p, 0, (ln * 8 + 1) * 4, stream)); BLD_TRY(hipMemsetAsync(own.p, 0, (ln + 1) * 4, stream));
        BLD_TRY(hipMemsetAsync(placed.p, 0, ln * 4, stream)); BLD_TRY(hipMemsetAsync(down.p, 0, ln * 4, stream)); BLD_TRY(hipMemsetAsync(kids.p, 0, (ln + 1) * 4, stream));
        BLD_TRY(hipMemsetAsync(fill.p, 0, ln * 8 * 4, stream)); BLD_TRY(hipMemsetAsync(fill_own.p, 0, ln * 4, stream));
        if (members)
            hipLaunchKernelGGL(k_bld_classify, dim3(blocks_for(members)), dim3(kThreads), 0, stream, P, pad, first_node, members, mem_prim[cur].p, mem_node[cur].p, mask.p, N,
                               pbox.p, skip.p, raw.p, d_small, off_obj, prim_obj.p, cnt.p, own.p, placed.p, down.p);
        hipLaunchKernelGGL(k_bld_decide, dim3(blocks_for(ln)), dim3(kThreads), 0, stream, first_node, level_nodes, N, cnt.p, own.p, placed.p, down.p, kids.p, failed.p);
        // exclusive scans over ln + 1 / ln * 8 + 1 entries: the last output is the total
        const size_t scan_items[3] = { ln + 1, ln * 8 + 1, ln + 1 };
        uint32_t* scan_in[3] = { kids.p, cnt.p, own.p };
        uint32_t* scan_out[3] = { kid_off.p, mem_off.p, own_off.p };
        for (int k = 0; k < 3; ++k) {
            size_t need = 0;
            BLD_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, scan_in[k], scan_out[k], (int)scan_items[k], stream));
            if (need > scan_tmp_bytes) { if (scan_tmp.p) { BLD_TRY(hipStreamSynchronize(stream)); (void)hipFree(scan_tmp.p); scan_tmp.p = nullptr; } BLD_TRY(scan_tmp.alloc(need)); scan_tmp_bytes = need; }
            BLD_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, need, scan_in[k], scan_out[k], (int)scan_items[k], stream));
        }
        uint32_t tot[3] = { 0, 0, 0 };
        BLD_TRY(hipMemcpyAsync(&tot[0], kid_off.p + ln, 4, hipMemcpyDeviceToHost, stream));
        BLD_TRY(hipMemcpyAsync(&tot[1], mem_off.p + ln * 8, 4, hipMemcpyDeviceToHost, stream));
        BLD_TRY(hipMemcpyAsync(&tot[2], own_off.p + ln, 4, hipMemcpyDeviceToHost, stream));
        BLD_TRY(hipStreamSynchronize(stream));
        const uint32_t next_nodes = tot[0], next_members = tot[1], level_refs = tot[2];
        if ((size_t)total_nodes + next_nodes >= (1u << 24)) { err = "device octree build: more than 2^24 nodes"; return 1; }
        if ((size_t)total_nodes + next_nodes > cap) {
            const size_t new_cap = std::max(cap * 2, (size_t)total_nodes + next_nodes + 4096);
            auto grow = [&](auto& buf, size_t per_node) -> hipError_t {
                typename std::remove_reference<decltype(buf)>::type bigger;
                hipError_t e = bigger.alloc(new_cap * per_node);
                if (e != hipSuccess) return e;
                e = hipMemcpyAsync(bigger.p, buf.p, (size_t)total_nodes * per_node * sizeof(*buf.p), hipMemcpyDeviceToDevice, stream);
                if (e != hipSuccess) return e;
                e = hipStreamSynchronize(stream);
                if (e != hipSuccess) return e;
                std::swap(buf.p, bigger.p); std::swap(buf.n, bigger.n);      // `bigger` frees the old array
                return hipSuccess;
            };
            BLD_TRY(grow(cell, 6)); BLD_TRY(grow(tight, 6)); BLD_TRY(grow(depth, 1)); BLD_TRY(grow(begin, 1)); BLD_TRY(grow(count, 1));
            BLD_TRY(grow(own_begin, 1)); BLD_TRY(grow(own_count, 1)); BLD_TRY(grow(child_word, 1));
            N = Nodes{ cell.p, tight.p, depth.p, begin.p, count.p, own_begin.p, own_count.p, child_word.p };
            cap = new_cap;
        }
        const int nxt = cur ^ 1;
        if (mem_prim[nxt].p) { (void)hipFree(mem_prim[nxt].p); mem_prim[nxt].p = nullptr; (void)hipFree(mem_node[nxt].p); mem_node[nxt].p = nullptr; }
        BLD_TRY(mem_prim[nxt].alloc(next_members)); BLD_TRY(mem_node[nxt].alloc(next_members));
        Buf<uint32_t>* refs = new Buf<uint32_t>();
        ref_parts.push_back(refs); ref_counts.push_back(level_refs);
        BLD_TRY(refs->alloc(level_refs));
        // own_begin is an index into THIS level's part; the global index (total_refs + ...) is what the records carry
        hipLaunchKernelGGL(k_bld_children, dim3(blocks_for(ln)), dim3(kThreads), 0, stream, first_node, level_nodes, total_nodes, total_refs, N, cnt.p, kid_off.p, mem_off.p, own_off.p);
        if (members)
            hipLaunchKernelGGL(k_bld_scatter, dim3(blocks_for(members)), dim3(kThreads), 0, stream, first_node, members, mem_prim[cur].p, mem_node[cur].p, mask.p, N, failed.p, fill.p,
                               fill_own.p, mem_prim[nxt].p, mem_node[nxt].p, refs->p, total_refs);
        BLD_TRY(hipGetLastError());
        BLD_TRY(hipStreamSynchronize(stream));       // the level's temporaries go out of scope
        first_node = total_nodes; total_nodes += next_nodes; level_nodes = next_nodes; total_refs += level_refs; members = next_members; cur = nxt; ++level;
    }

    // ---- the BIG image: nodes, leaf records (+ pad record), then the host's triangle and pair records
    R.nnode = total_nodes; R.nrefs = total_refs; R.depth = deepest;
    uint64_t off = 0;
    R.off_node = 0; off += (uint64_t)total_nodes * NODE_STRIDE;
    R.off_leaf = (uint32_t)off; off += ((uint64_t)total_refs + 1u) * LEAF_STRIDE;
    R.off_tri = (uint32_t)off; R.off_pair = (uint32_t)(off + tail_off_pair);
    off += tail.size();
    if (off >= 0xFFFFFFF0ull) { err = "scene image exceeds 16 GiB of words"; return 1; }
    R.big_words = off;
    BLD_TRY(hipMalloc(&R.d_big, std::max<uint64_t>(off, 1) * 4));
    if (!tail.empty()) BLD_TRY(hipMemcpyAsync(R.d_big + R.off_tri, tail.data(), tail.size() * 4, hipMemcpyHostToDevice, stream));
    Buf<uint32_t> refs_all;
    BLD_TRY(refs_all.alloc(total_refs));
    {
        uint32_t at = 0;
        for (size_t k = 0; k < ref_parts.size(); ++k) {
            if (ref_counts[k]) BLD_TRY(hipMemcpyAsync(refs_all.p + at, ref_parts[k]->p, (size_t)ref_counts[k] * 4, hipMemcpyDeviceToDevice, stream));
            at += ref_counts[k];
        }
    }
    hipLaunchKernelGGL(k_bld_emit_nodes, dim3(blocks_for(total_nodes)), dim3(kThreads), 0, stream, total_nodes, N, R.d_big, R.off_node);
    hipLaunchKernelGGL(k_bld_emit_leaves, dim3(blocks_for((uint64_t)total_refs + 1)), dim3(kThreads), 0, stream, total_refs, refs_all.p, prim_obj.p, d_small, off_obj, R.d_big,
                       R.off_leaf, R.off_tri);
    BLD_TRY(hipGetLastError());
    BLD_TRY(hipEventRecord(ev1, stream));
    BLD_TRY(hipStreamSynchronize(stream));
    BLD_TRY(hipEventElapsedTime(&R.build_ms, ev0, ev1));
    for (int a = 0; a < 6; ++a) R.root_cell[a] = rootinfo_h[a];
    R.pad = pad;
    return 0;
}

// ---------------------------------------------------------------- refit (second half)
// A scene whose primitives only MOVED since the tree was built - same objects, same triangle counts - keeps its tree when every moved
// primitive is still referenced wherever the builder's rule wants a reference: at every node on the way down, the node's own list holds
// it, or every child whose padded cell the primitive (padded box, then the triangle itself: the builder's two tests) now overlaps
// exists and passes the same check.  Extra references do no harm (a test that misses), a missing one would hide a hit - then the
// caller rebuilds.  What is recomputed: the inlined triangle data of the leaf records and every node's tight box, bottom-up.
// The reference's counterpart is Octree::InsertIfOverlaps (source/accelerationstructures/octree.h:50-70), its incremental entry point.

// a node's cell from its record (min corner + 2^-depth): the builder's max corner to within an ulp; users add `slack`
__device__ __forceinline__ void rf_cell(const uint32_t* w, const float* root_size, float* cmn, float* cmx)
{
    const float scale = __uint_as_float(w[NODE_SCALE] & NODE_SCALE_MASK);
    for (int a = 0; a < 3; ++a) { cmn[a] = __uint_as_float(w[NODE_CELL + a]); cmx[a] = cmn[a] + root_size[a] * scale; }
}

struct RefitArgs {
    const uint32_t* big;          // records (read) ...
    uint32_t off_node, off_leaf;
    float root_size[3], root_min[3], root_max[3], pad;
};

// The builder's cell of the node reached from the root by `path` (3 bits per level, the root's choice in the lowest bits): the
// float formulas of k_bld_children / Builder::make applied level by level, so the corners are the builder's own bit for bit.
__device__ __forceinline__ void rf_cell_by_path(const float* root_min, const float* root_max, uint64_t path, uint32_t depth, float* cmn, float* cmx)
{
    for (int a = 0; a < 3; ++a) { cmn[a] = root_min[a]; cmx[a] = root_max[a]; }
    for (uint32_t l = 0; l < depth; ++l) {
        const uint32_t c = (uint32_t)(path >> (3u * l)) & 7u;
        for (int a = 0; a < 3; ++a) {
            const float half = 0.5f * (cmx[a] - cmn[a]), centre = cmn[a] + half;
            if ((c >> a) & 1u) cmn[a] = centre; else cmx[a] = centre;
        }
    }
}

// one thread per primitive of a moved object: is it still referenced wherever the BUILDER's rule (k_bld_classify: padded-box overlap
// with the child cell, then the triangle against the padded child cell) wants a reference?  fail[0] != 0: no
__global__ void k_rf_check(RefitArgs A, Params P, uint32_t nprim, const uint8_t* __restrict__ moved_obj, const uint32_t* __restrict__ prim_obj, const uint8_t* __restrict__ prim_skip,
                           const float* __restrict__ pbox, const float* __restrict__ raw_tris, const uint32_t* __restrict__ small_img, uint32_t off_obj, uint32_t* __restrict__ fail)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprim) return;
    const uint32_t obj = prim_obj[p];
    if (!moved_obj[obj] || prim_skip[p]) return;
    const float pad = A.pad;
    float bmn[3], bmx[3];
    for (int a = 0; a < 3; ++a) { bmn[a] = pbox[(size_t)p * 6 + a] - pad; bmx[a] = pbox[(size_t)p * 6 + 3 + a] + pad; }
    // inside the root cell (the builder's root cell holds every padded box with a padding to spare)
    for (int a = 0; a < 3; ++a)
        if (!(bmn[a] >= A.root_min[a] && bmx[a] <= A.root_max[a])) { atomicOr(fail, 1u); return; }
    const uint32_t* o = small_img + off_obj + obj * OBJ_STRIDE;
    const bool tri = o[OBJ_KIND] != (uint32_t)MCRT_OBJECT_SPHERE;
    const float* t = tri ? raw_tris + (size_t)(o[OBJ_FIRST_TRI] + (p - o[OBJ_FIRST_PRIM])) * 12 : nullptr;
    constexpr int kStack = 8 * MAX_TREE_DEPTH + 8;
    uint32_t stack[kStack];         // node index | depth << 24
    uint64_t paths[kStack];
    int sp = 0;
    stack[0] = 0u; paths[0] = 0ull; sp = 1;
    while (sp > 0) {
        --sp;
        const uint32_t node = stack[sp] & 0x00FFFFFFu, depth = stack[sp] >> 24;
        const uint64_t path = paths[sp];
        const uint32_t* w = A.big + A.off_node + (size_t)node * NODE_STRIDE;
        bool own = false;
        if (w[NODE_LEAF] != NO_REFS) {
            for (uint32_t r = w[NODE_LEAF];; ++r) {
                const uint32_t e = A.big[A.off_leaf + (size_t)r * LEAF_STRIDE + LEAF_PRIM];
                if ((e & ~REF_LAST) == p) { own = true; break; }
                if (e & REF_LAST) break;
            }
        }
        if (own) continue;
        const uint32_t cw = w[NODE_CHILD], mask = cw >> 24, base = cw & 0x00FFFFFFu;
        if (!mask) { atomicOr(fail, 2u); return; }
        float cmn[3], cmx[3], centre[3];
        rf_cell_by_path(A.root_min, A.root_max, path, depth, cmn, cmx);
        for (int a = 0; a < 3; ++a) centre[a] = cmn[a] + 0.5f * (cmx[a] - cmn[a]);
        for (uint32_t c = 0; c < 8; ++c) {
            float kmn[3], kmx[3];
            bool over = true;
            for (int a = 0; a < 3; ++a) {
                const bool hi = (c >> a) & 1u;
                kmn[a] = hi ? centre[a] : cmn[a];
                kmx[a] = hi ? cmx[a] : centre[a];
                if (bmx[a] < kmn[a] || kmx[a] < bmn[a]) over = false;
            }
            if (!over) continue;
            if (tri) {
                float pmn[3], pmx[3];
                for (int a = 0; a < 3; ++a) { pmn[a] = kmn[a] - pad; pmx[a] = kmx[a] + pad; }
                if (!tri_overlaps_box(t, pmn, pmx)) continue;
            }
            if (!(mask & (1u << c))) { atomicOr(fail, 4u); return; }
            if (sp >= kStack || depth >= 255u) { atomicOr(fail, 8u); return; }
            stack[sp] = (base + (uint32_t)__popc(mask & ((1u << c) - 1u))) | ((depth + 1u) << 24);
            paths[sp] = path | ((uint64_t)c << (3u * depth));
            ++sp;
        }
    }
}

// leaf records of moved meshes: v0 / edge1 / edge2 again from the (re-uploaded) triangle records
__global__ void k_rf_leaves(uint32_t nrefs, uint32_t* __restrict__ big, uint32_t off_leaf, uint32_t off_tri, const uint8_t* __restrict__ moved_obj,
                            const uint32_t* __restrict__ small_img, uint32_t off_obj)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrefs) return;
    uint32_t* w = big + off_leaf + (size_t)r * LEAF_STRIDE;
    const uint32_t ow = w[LEAF_OBJ];
    if (ow & LEAF_SPHERE) return;
    if (!moved_obj[ow]) return;
    const uint32_t* o = small_img + off_obj + ow * OBJ_STRIDE;
    const uint32_t prim = w[LEAF_PRIM] & ~REF_LAST;
    const uint32_t* tr = big + off_tri + (size_t)(o[OBJ_FIRST_TRI] + (prim - o[OBJ_FIRST_PRIM])) * TRI_STRIDE;
    for (uint32_t k = 0; k < 9; ++k) w[k] = tr[k];
}

// per node: the union of its OWN members' padded boxes clipped to the padded cell -> box[node][6] (min xyz, max xyz; empty: +max / -max)
__global__ void k_rf_own(RefitArgs A, uint32_t nodes, const float* __restrict__ pbox, float* __restrict__ box)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nodes) return;
    const uint32_t* w = A.big + A.off_node + (size_t)n * NODE_STRIDE;
    float mn[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f }, mx[3] = { -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f };
    if (w[NODE_LEAF] != NO_REFS) {
        float cmn[3], cmx[3];
        rf_cell(w, A.root_size, cmn, cmx);
        const float grow = A.pad + A.pad * (1.0f / 64.0f);
        for (uint32_t r = w[NODE_LEAF];; ++r) {
            const uint32_t e = A.big[A.off_leaf + (size_t)r * LEAF_STRIDE + LEAF_PRIM], p = e & ~REF_LAST;
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], fmaxf(pbox[(size_t)p * 6 + a] - A.pad, cmn[a] - grow));
                mx[a] = fmaxf(mx[a], fminf(pbox[(size_t)p * 6 + 3 + a] + A.pad, cmx[a] + grow));
            }
            if (e & REF_LAST) break;
        }
    }
    for (int a = 0; a < 3; ++a) { box[(size_t)n * 6 + a] = mn[a]; box[(size_t)n * 6 + 3 + a] = mx[a]; }
}

// one sweep of box[node] = box[node] U box[children]: after (tree depth) sweeps every box holds its whole subtree (boxes only grow, so
// a sweep that reads a child mid-update is harmless)
__global__ void k_rf_up(RefitArgs A, uint32_t nodes, float* __restrict__ box)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nodes) return;
    const uint32_t cw = A.big[A.off_node + (size_t)n * NODE_STRIDE + NODE_CHILD], mask = cw >> 24, base = cw & 0x00FFFFFFu;
    if (!mask) return;
    float b[6];
    for (int k = 0; k < 6; ++k) b[k] = box[(size_t)n * 6 + k];
    const uint32_t kids = (uint32_t)__popc(mask);
    for (uint32_t c = 0; c < kids; ++c)
        for (int a = 0; a < 3; ++a) {
            b[a] = fminf(b[a], box[(size_t)(base + c) * 6 + a]);
            b[3 + a] = fmaxf(b[3 + a], box[(size_t)(base + c) * 6 + 3 + a]);
        }
    for (int k = 0; k < 6; ++k) box[(size_t)n * 6 + k] = b[k];
}

__global__ void k_rf_store(uint32_t nodes, const float* __restrict__ box, uint32_t* __restrict__ big, uint32_t off_node)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nodes) return;
    if (box[(size_t)n * 6] > box[(size_t)n * 6 + 3]) return;           // nothing below (the root of an empty scene): the record keeps its box
    uint32_t* w = big + off_node + (size_t)n * NODE_STRIDE;
    for (int k = 0; k < 6; ++k) w[NODE_BOX + k] = __float_as_uint(box[(size_t)n * 6 + k]);
}

// Returns 0 (refitted), 2 (the tree no longer fits: `why` holds the check's verdict, nothing usable changed - rebuild) or 1 with `err` set.
// d_big already holds the NEW triangle / pair records at off_tri; d_small the new SMALL image.
inline int refit_on_device(hipStream_t stream, const Params& P, const uint32_t* d_small, uint32_t off_obj, const float* h_raw_tris, const std::vector<uint8_t>& moved,
                           uint32_t* d_big, uint32_t off_node, uint32_t off_leaf, uint32_t off_tri, uint32_t nnode, uint32_t nrefs, uint32_t depth, const float root_cell[6],
                           float pad, uint32_t& why, float& kernel_ms, std::string& err)
{
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } events;
    BLD_TRY(hipEventCreate(&events.a)); BLD_TRY(hipEventCreate(&events.b));
    Buf<float> raw, pbox, box;
    Buf<uint32_t> prim_obj, fail;
    Buf<uint8_t> skip, moved_d;
    Buf<int> rootbox;
    BLD_TRY(raw.alloc((size_t)P.ntri * 12)); BLD_TRY(pbox.alloc((size_t)P.nprim * 6)); BLD_TRY(prim_obj.alloc(P.nprim)); BLD_TRY(skip.alloc(P.nprim));
    BLD_TRY(rootbox.alloc(8)); BLD_TRY(fail.alloc(1)); BLD_TRY(moved_d.alloc(moved.size())); BLD_TRY(box.alloc((size_t)nnode * 6));
    if (P.ntri) BLD_TRY(hipMemcpyAsync(raw.p, h_raw_tris, (size_t)P.ntri * 48, hipMemcpyHostToDevice, stream));
    BLD_TRY(hipMemcpyAsync(moved_d.p, moved.data(), moved.size(), hipMemcpyHostToDevice, stream));
    BLD_TRY(hipEventRecord(events.a, stream));
    BLD_TRY(hipMemsetAsync(fail.p, 0, 4, stream));
    const int init_box[8] = { 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0 };
    BLD_TRY(hipMemcpyAsync(rootbox.p, init_box, sizeof init_box, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_bld_boxes, dim3(blocks_for(P.nprim)), dim3(kThreads), 0, stream, P, d_small, off_obj, raw.p, pbox.p, prim_obj.p, skip.p, rootbox.p);
    RefitArgs A;
    A.big = d_big; A.off_node = off_node; A.off_leaf = off_leaf; A.pad = pad;
    for (int a = 0; a < 3; ++a) { A.root_min[a] = root_cell[a]; A.root_max[a] = root_cell[3 + a]; A.root_size[a] = root_cell[3 + a] - root_cell[a]; }
    hipLaunchKernelGGL(k_rf_check, dim3(blocks_for(P.nprim)), dim3(kThreads), 0, stream, A, P, P.nprim, moved_d.p, prim_obj.p, skip.p, pbox.p, raw.p, d_small, off_obj, fail.p);
    BLD_TRY(hipGetLastError());
    BLD_TRY(hipMemcpyAsync(&why, fail.p, 4, hipMemcpyDeviceToHost, stream));
    BLD_TRY(hipStreamSynchronize(stream));
    if (why) return 2;
    if (nrefs) hipLaunchKernelGGL(k_rf_leaves, dim3(blocks_for(nrefs)), dim3(kThreads), 0, stream, nrefs, d_big, off_leaf, off_tri, moved_d.p, d_small, off_obj);
    hipLaunchKernelGGL(k_rf_own, dim3(blocks_for(nnode)), dim3(kThreads), 0, stream, A, nnode, pbox.p, box.p);
    for (uint32_t s = 0; s <= depth; ++s) hipLaunchKernelGGL(k_rf_up, dim3(blocks_for(nnode)), dim3(kThreads), 0, stream, A, nnode, box.p);
    hipLaunchKernelGGL(k_rf_store, dim3(blocks_for(nnode)), dim3(kThreads), 0, stream, nnode, box.p, d_big, off_node);
    BLD_TRY(hipGetLastError());
    BLD_TRY(hipEventRecord(events.b, stream));
    BLD_TRY(hipStreamSynchronize(stream));
    BLD_TRY(hipEventElapsedTime(&kernel_ms, events.a, events.b));
    return 0;
}

} // namespace bld
} // namespace mcrt
