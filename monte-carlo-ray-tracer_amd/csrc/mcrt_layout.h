// mcrt_layout.h - packed scene image shared by the host flattener and the kernels.
//
// Two word-addressed images are built by mcrt_upload_scene:
//   SMALL: per-object records (geometry header, mesh gate box, light vectors,
//          material), the light list and the background colour.  A few KB at
//          most; every kernel stages it in LDS.
//   BIG:   the flat octree (48-byte node records + per-node lists of 48-byte
//          leaf records that carry the triangle they refer to), the triangle
//          records and the triangle-pair records of the scan path.
//          Staged in LDS when SMALL+BIG fit the LDS budget (all Cornell-box
//          configurations), read from HBM/L2 otherwise (the 1M-triangle mesh).
//
// Reference types being flattened: Object/SphereObject/TriangleMesh/Light
// (source/objects/*.h), Material (source/core/material.h:11-19), Triangle
// (source/core/triangle.h:8-23), Octree (source/accelerationstructures/octree.h).
#pragma once
#include <stdint.h>

namespace mcrt {

// ---- object record (SMALL image), OBJ_STRIDE words each -------------------
enum : uint32_t {
    OBJ_KIND = 0,        // MCRT_OBJECT_*
    OBJ_FIRST_TRI = 1,
    OBJ_TRI_COUNT = 2,
    OBJ_FIRST_PRIM = 3,  // id of this object's first primitive (global (object, element) order)
    OBJ_RADIUS = 4,
    OBJ_POS = 5,         // 3 floats
    OBJ_AABB_MIN = 8,    // 3 floats: per-mesh gate, source/objects/mesh.cpp:11-16
    OBJ_AABB_MAX = 11,
    OBJ_LIGHT_X = 14,    // Light::xVector
    OBJ_LIGHT_Y = 17,    // Light::yVector
    OBJ_SURFACE = 20,
    OBJ_DIFFUSE = 21,
    OBJ_ALBEDO = 22,
    OBJ_ROUGHNESS = 23,
    OBJ_IOR = 24,
    OBJ_FIRST_PAIR = 25, // index of this mesh's first triangle-pair record (scan path)
    OBJ_COLOR = 26,      // 3 doubles (8-byte aligned: stride and offset are even)
    OBJ_EMISSION = 32,   // 3 doubles
    OBJ_ON_A = 38,       // Oren-Nayar A and B of source/core/material.h:29-33: functions of the roughness alone,
    OBJ_ON_B = 39,       // evaluated once at upload with the reference's float operations
    OBJ_STRIDE = 40
};

// ---- triangle record (BIG image), TRI_STRIDE words each, 16-byte aligned --
// v0, edge1 = v1 - v0, edge2 = v2 - v0 (the two subtractions of
// source/core/triangle.h:31-32 done once at upload: same floats), face normal.
enum : uint32_t { TRI_V0 = 0, TRI_E1 = 3, TRI_E2 = 6, TRI_N = 9, TRI_STRIDE = 12 };

// ---- triangle-pair record (BIG image, scan path), PAIR_STRIDE words each, 16-byte aligned.
// Triangles 2j and 2j+1 of a mesh interleaved component by component -
// v0x[a,b] v0y[a,b] v0z[a,b] e1x[a,b] ... e2z[a,b] - so that one scalar load brings SGPR pairs
// that feed packed (2-wide) float instructions directly.  An odd last triangle is paired with a
// copy of itself (a repeated test can never win the strict `<`).
enum : uint32_t { PAIR_V0 = 0, PAIR_E1 = 6, PAIR_E2 = 12, PAIR_STRIDE = 20 };

// ---- octree node record (BIG image), NODE_STRIDE words each, 16-byte aligned: three 16-byte loads bring everything a
// visit needs - the node's own tight box, its child word and leaf list, and its CELL (the octant it owns: min corner and
// 2^-depth, so that cell size = root size * scale).  The children's boxes are NOT fetched to decide where a ray goes
// next: which octants of the cell the ray passes follows from the cell's three mid planes (OctreeWalk::visit), and a
// child's record is read only when the child is popped.
enum : uint32_t {
    NODE_BOX = 0,        // min.xyz, max.xyz (tight, padded)
    NODE_CHILD = 6,      // bits 0..23: index of the first child (children are contiguous,
                         // only non-empty octants are stored), bits 24..31: octant mask
    NODE_LEAF = 7,       // index of this node's first leaf record, NO_REFS if none
    NODE_CELL = 8,       // min corner of the node's cell, 3 floats (the builder's own values)
    NODE_SCALE = 11,     // 2^-depth as a float (sign and exponent bits, the mantissa of a power of two is zero) with the LENGTH of
                         // the node's own list in the 23 mantissa bits: a lane that begins a list knows how many records follow
#ifdef MCRT_NODE_STRIDE
    NODE_STRIDE = MCRT_NODE_STRIDE
#else
    NODE_STRIDE = 12
#endif
};
constexpr uint32_t NODE_SCALE_MASK = 0xFF800000u, NODE_COUNT_MASK = 0x007FFFFFu;
constexpr uint32_t MAX_PRIMITIVES = NODE_COUNT_MASK;      // a list holds a primitive once at most
// ---- leaf record (BIG image), LEAF_STRIDE words each, 16-byte aligned.  One per (node, primitive)
// reference, a node's records contiguous: the triangle's v0/edge1/edge2 are INLINED, so walking a
// node's list is one sequential stream of 48-byte records (no ref -> object -> triangle pointer
// chase).  A primitive referenced from several nodes is stored once per reference: HBM is plentiful.
enum : uint32_t {
    LEAF_V0 = 0, LEAF_E1 = 3, LEAF_E2 = 6,
    LEAF_PRIM = 9,       // primitive id, REF_LAST set on the last record of the node's list
    LEAF_OBJ = 10,       // object index, LEAF_SPHERE set for a sphere (geometry read from the object record)
    LEAF_STRIDE = 12
};
constexpr uint32_t LEAF_SPHERE = 0x80000000u;
constexpr uint32_t LEAF_RESUME = 0x80000000u;  // traversal stack only: the entry is not a node index but a leaf list to resume part-way (record index in the low bits)
constexpr uint32_t NO_REFS = 0xFFFFFFFFu;
constexpr uint32_t REF_LAST = 0x80000000u;   // set on the last reference of a node's list
constexpr uint32_t MISS = 0xFFFFFFFFu;

// The walk's margins against its own rounding.  The builder pads cells and boxes by `pad` (mcrt_flatten.cpp); the walk routes a ray
// with walk_pad = kWalkPadFraction * pad, which leaves pad / 2 between "the walk treats the ray as within reach of a child" and "the
// builder referenced the primitive from that child".  The walk's plane and slab distances are off by a few ulps of max(|origin|,
// |cell|) in position units; pad >= 256 ulps of the largest scene coordinate, so the margin holds while the origin is within
// far_limit = pad * kFarOriginFactor = pad / (32 * FLT_EPSILON), i.e. while 16 ulps of the origin's coordinate stay below pad / 2.
constexpr float kWalkPadFraction = 0.5f;
constexpr float kFarOriginFactor = 262144.0f;      // 2^18 = 1 / (32 * FLT_EPSILON)
constexpr uint32_t MAX_TREE_DEPTH = 12;
constexpr uint32_t TRAVERSAL_STACK = 7 * MAX_TREE_DEPTH + 4;
constexpr uint32_t MAX_OBJECTS_GATED = 64;   // the per-ray mesh-gate cache is a 64-bit mask: objects 0..63 are cached, later ones evaluated on every contact
constexpr uint32_t MAX_OBJECTS = 512;        // object table (160 bytes each) has to fit the LDS budget next to the walk's stack

struct SceneView {
    const uint32_t* small_img;   // global copies of the two images
    const uint32_t* big_img;
    uint32_t small_words, big_words;
    uint32_t nobj, nlights, ntri, nprim, nnode;
    uint32_t off_obj, off_lights, off_bg;            // into SMALL
    uint32_t off_node, off_leaf, off_tri, off_pair;  // into BIG (all multiples of 4 words)
    uint32_t has_pairs;          // triangle-pair records present (scan path usable)
    uint32_t tri_src;            // word offset of the triangle records in the global BIG image (== off_tri except in the
                                 // view handed to the flat-scan kernels, whose off_tri addresses their LDS copy)
    float root_size[3];          // extent of the root cell (a node's cell size = root_size * its NODE_SCALE)
    float walk_pad;              // HALF the builder's padding: the walk decides "within the padding of a mid plane" with this, so that
                                 // its own rounding (about an ulp of the origin's coordinate) can never put a ray into a child the
                                 // builder did not reference the primitive from (kWalkPadFraction, OctreeWalk::visit)
    float far_limit;             // origins with a coordinate beyond this (or NaN) are answered by the loop over all objects: there an
                                 // ulp of the coordinate is no longer small against walk_pad (kFarOriginFactor, OctreeWalk::begin)
};

} // namespace mcrt
