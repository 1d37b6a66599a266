// mcrt_scene.cpp - host-side scene API (see mcrt_scene.hpp).  Scene
// construction reproduces the reference's float arithmetic so that the
// flattened geometry is bit-identical to what its own classes hold; ray
// arithmetic lives in libmcrt_hip.so.  Build with -ffp-contract=off.
#include "mcrt_scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>

// ------------------------------------------------------------------ geometry
// AABB::Encapsulate, source/core/aabb.h:98-107
void AABB::Encapsulate(const vec3& p)
{
    min.x = std::min(min.x, p.x); min.y = std::min(min.y, p.y); min.z = std::min(min.z, p.z);
    max.x = std::max(max.x, p.x); max.y = std::max(max.y, p.y); max.z = std::max(max.z, p.z);
}

// Triangle ctor, source/core/triangle.h:17-23
Triangle::Triangle(const vec3& v0, const vec3& v1, const vec3& v2) : vertex0(v0), vertex1(v1), vertex2(v2)
{
    normal = normalize(cross(v1 - v0, v2 - v0));
}

// source/core/randomization.cpp:8-13
UniformRandomGenerator::UniformRandomGenerator()
{
    std::random_device rd;
    xorseed[0] = (uint64_t(rd()) << 32) ^ (rd());
    xorseed[1] = (uint64_t(rd()) << 32) ^ (rd());
}
// RandomInt + to_double, source/core/randomization.h:23-43
double UniformRandomGenerator::RandomDouble()
{
    uint64_t x = xorseed[0];
    const uint64_t y = xorseed[1];
    xorseed[0] = y;
    x ^= x << 23;
    xorseed[1] = x ^ y ^ (x >> 17) ^ (y >> 26);
    const uint64_t bits = (0x3FFull << 52) | ((xorseed[1] + y) >> 12);
    double d;
    std::memcpy(&d, &bits, sizeof d);
    return d - 1.0;
}
UniformRandomGenerator UniformRandomGenerator::ForSample(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t st[2];
    mcrt_rng_seed(seed, pixel, sample, st);
    return UniformRandomGenerator(st[0], st[1]);
}

// TriangleMesh::AddQuad, source/objects/mesh.cpp:49-53
void TriangleMesh::AddQuad(vec3 p1, vec3 p2, vec3 p3, vec3 p4)
{
    triangles.push_back(Triangle{ p1, p2, p3 });
    triangles.push_back(Triangle{ p3, p4, p1 });
}

// TriangleMesh::UpdateAABB, source/objects/mesh.cpp:55-68: starts from the degenerate box at `position`
void TriangleMesh::UpdateAABB()
{
    if (triangles.empty()) return;
    aabb = AABB(position, vec3{ 0.0f });
    for (const Triangle& t : triangles) {
        aabb.Encapsulate(t.vertex0);
        aabb.Encapsulate(t.vertex1);
        aabb.Encapsulate(t.vertex2);
    }
}

// Replaces TriangleMesh::LoadMesh (source/objects/mesh.cpp:70-105), which ignores the face indices
// (SURVEY.md 0.4).  Reads `v` and `f` records; faces are fan-triangulated (a,b,c),(a,c,d),... in file
// order; vertices are offset by `position` as the reference does.  Prints and returns false on error.
// ---- polygon triangulation as the reference's OBJ reader does it (source/helpers/OBJ_Loader.h:820-985, "OBJL": the library
// TriangleMesh::LoadMesh reads files with, source/objects/mesh.cpp:70-80), so that a face with more than three corners becomes
// the triangles the reference's loader would get.  The procedure, quirks included (pinned by tests/golden/obj_triangulation.npz,
// which OBJL itself produced):
//   - ear clipping from the front of the corner list: the first corner whose triangle (previous, corner, next) holds no other
//     corner of the face is cut off, and the search starts over; with four corners left the face is closed with (previous,
//     corner, next) and (previous, next, the fourth);
//   - a triangle's corners are always emitted in ASCENDING position within the face (OBJL matches them back by position);
//   - the containment test only works for faces whose plane passes through the coordinate origin: OBJL projects the corner's
//     POSITION VECTOR onto the triangle's normal and asks for an exact zero (OBJ_Loader.h:291-301).  Everywhere else nothing is
//     ever "inside", and an n-gon comes out as (k, k+1, n-1), k = 0 .. n-3 - for a quad (0,1,3), (1,2,3);
//   - its angle test (`angle <= 0 && angle >= 180`, :931-933) can never hold and is left out.
namespace {
struct P3 { float x, y, z; };
inline bool same(const P3& a, const P3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline P3 sub(const P3& a, const P3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline P3 crossp(const P3& a, const P3& b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float dotp(const P3& a, const P3& b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline float length(const P3& a) { return sqrtf(powf(a.x, 2) + powf(a.y, 2) + powf(a.z, 2)); }      // OBJL's MagnitudeV3, :217-220
// both on the same side of the line a-b (OBJ_Loader.h:257-266)
inline bool sameSide(const P3& p1, const P3& p2, const P3& a, const P3& b) { return dotp(crossp(sub(b, a), sub(p1, a)), crossp(sub(b, a), sub(p2, a))) >= 0; }
// OBJ_Loader.h:280-302
bool inTriangle(const P3& point, const P3& t1, const P3& t2, const P3& t3)
{
    if (!(sameSide(point, t1, t2, t3) && sameSide(point, t2, t1, t3) && sameSide(point, t3, t1, t2))) return false;
    const P3 n = crossp(sub(t2, t1), sub(t3, t1));
    const float len = length(n);
    const P3 bn = { n.x / len, n.y / len, n.z / len };
    const float along = dotp(point, bn);
    const P3 proj = { bn.x * along, bn.y * along, bn.z * along };
    return length(proj) == 0;
}
// every corner of `face` whose position equals one of the three, in face order
void emit(const std::vector<P3>& face, size_t upTo, const P3& a, const P3& b, const P3& c, std::vector<unsigned>& out)
{
    for (size_t j = 0; j < upTo; ++j) {
        if (same(face[j], a)) out.push_back(unsigned(j));
        if (same(face[j], b)) out.push_back(unsigned(j));
        if (same(face[j], c)) out.push_back(unsigned(j));
    }
}
void triangulate(const std::vector<P3>& face, std::vector<unsigned>& out)
{
    if (face.size() < 3) return;
    if (face.size() == 3) { out.push_back(0); out.push_back(1); out.push_back(2); return; }
    std::vector<P3> left = face;
    for (;;) {
        for (int i = 0; i < int(left.size()); ++i) {
            const P3 prev = left[i == 0 ? left.size() - 1 : size_t(i) - 1], cur = left[size_t(i)], next = left[size_t(i) + 1 == left.size() ? 0 : size_t(i) + 1];
            if (left.size() == 3) {
                emit(face, left.size(), cur, prev, next, out);           // (OBJL scans only the first three corners of the face here, :876-884)
                left.clear();
                break;
            }
            if (left.size() == 4) {
                emit(face, face.size(), cur, prev, next, out);
                P3 fourth = { 0, 0, 0 };
                for (const P3& q : left)
                    if (!same(q, cur) && !same(q, prev) && !same(q, next)) { fourth = q; break; }
                emit(face, face.size(), prev, next, fourth, out);
                left.clear();
                break;
            }
            bool holds = false;
            for (const P3& q : face)
                if (inTriangle(q, prev, cur, next) && !same(q, prev) && !same(q, cur) && !same(q, next)) { holds = true; break; }
            if (holds) continue;
            emit(face, face.size(), cur, prev, next, out);
            for (size_t j = 0; j < left.size(); ++j)
                if (same(left[j], cur)) { left.erase(left.begin() + long(j)); break; }
            i = -1;
        }
        if (out.empty() || left.empty()) break;
    }
}
} // namespace

// The reference's LoadMesh (source/objects/mesh.cpp:70-105) hands the file to OBJL and then builds its triangles from
// Vertices[k..k+2] in a sliding window, ignoring Indices and reading past the end.  This one builds them from
// the indices: v / f records (f corners as "v", "v/vt", "v/vt/vn", "v//vn", negative = relative to the end, as OBJL resolves
// them, :382-390), faces triangulated as OBJL triangulates them (above), every vertex offset by `position` (mesh.cpp:98-100).
bool TriangleMesh::LoadMesh(std::string path)
{
    std::ifstream in(path);
    if (!in) {
        std::printf("Failed to load mesh from path: %s\r\n", path.c_str());
        return false;
    }
    std::vector<P3> verts;
    std::string line;
    size_t before = triangles.size();
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            float x, y, z;
            if (ls >> x >> y >> z) verts.push_back({ x, y, z });
        } else if (tag == "f") {
            std::vector<P3> face;
            std::string tok;
            bool ok = true;
            while (ls >> tok) {
                long i = std::strtol(tok.c_str(), nullptr, 10);
                if (i < 0) i = long(verts.size()) + i + 1;               // relative indices
                if (i < 1 || i > long(verts.size())) { ok = false; break; }
                face.push_back(verts[size_t(i - 1)]);
            }
            if (!ok) continue;
            std::vector<unsigned> idx;
            triangulate(face, idx);
            for (size_t k = 0; k + 2 < idx.size(); k += 3) {
                const P3 &a = face[idx[k]], &b = face[idx[k + 1]], &c = face[idx[k + 2]];
                triangles.push_back(Triangle{ vec3(a.x, a.y, a.z) + position, vec3(b.x, b.y, b.z) + position, vec3(c.x, c.y, c.z) + position });
            }
        }
    }
    std::printf("Loaded obj: %s (%zu triangles)\r\n", path.c_str(), triangles.size() - before);
    return triangles.size() > before;
}

// Box::SetGeometry, source/objects/box.h:14-55
void Box::SetGeometry(vec3 basePosition, vec3 upVector, vec3 sideVector, float width, float depth, float height)
{
    position = basePosition;
    upVector = normalize(upVector);
    sideVector = normalize(sideVector);
    const vec3 localZ = cross(sideVector, upVector);
    const vec3 localX = cross(upVector, localZ);
    const float halfWidth = width / 2.0f, halfDepth = depth / 2.0f;

    vec3 base[4] = {
        localX * halfWidth + localZ * halfDepth,
        localX * halfWidth - localZ * halfDepth,
        localX * halfWidth * -1.0f - localZ * halfDepth,
        localX * halfWidth * -1.0f + localZ * halfDepth,
    };
    vec3 top[4];
    for (int i = 0; i < 4; ++i) top[i] = base[i] + upVector * height;
    for (int i = 0; i < 4; ++i) base[i] += basePosition;
    for (int i = 0; i < 4; ++i) top[i] += basePosition;

    AddQuad(base[3], base[2], base[1], base[0]);       // bottom
    AddQuad(top[0], top[1], top[2], top[3]);           // top
    for (int i = 0; i < 4; ++i) {                      // the four sides, in the reference's order
        const int j = (i + 1) & 3;
        AddQuad(base[i], base[j], top[j], top[i]);
    }
}

// Light::SetGeometry, source/objects/light.h:22-43
void Light::SetGeometry(vec3 centerPosition, vec3 lightDirection, vec3 sideDirection, vec2 quadDimensions)
{
    area = quadDimensions.x * quadDimensions.y;
    normal = normalize(lightDirection);
    position = centerPosition;
    yVector = normalize(cross(sideDirection, normal));
    xVector = normalize(cross(yVector, normal));
    xVector *= quadDimensions.x / 2.0f;
    yVector *= quadDimensions.y / 2.0f;
    const vec3 p1 = position - xVector - yVector;
    const vec3 p2 = position - xVector + yVector;
    const vec3 p3 = position + xVector + yVector;
    const vec3 p4 = position + xVector - yVector;
    position += lightDirection * INTERSECTION_ERROR_MARGIN;
    AddQuad(p1, p2, p3, p4);
}

// --------------------------------------------------------------- pixel buffer
// source/core/pixelbuffer.cpp:7-23
PixelBuffer::PixelBuffer(unsigned int width, unsigned int height) : imageWidth(width), imageHeight(height)
{
    dataSize = imageWidth * imageHeight * 3;
    data.assign(dataSize, 0.0);
    dx = 2.0 / double(imageWidth);
    dy = 2.0 / double(imageHeight);
    aspect = imageWidth / double(imageHeight);
}
void PixelBuffer::Accumulate(unsigned int x, unsigned int y, ColorDbl color)
{
    const unsigned int i = PixelArrayIndex(x, y);
    data[i] += color.r; data[i + 1] += color.g; data[i + 2] += color.b;
    if (rayCount.empty()) rayCount.assign(size_t(imageWidth) * imageHeight, 0);
    rayCount[size_t(y) * imageWidth + x]++;
}
ColorDbl PixelBuffer::GetPixelColor(unsigned int x, unsigned int y)
{
    const unsigned int i = PixelArrayIndex(x, y);
    return ColorDbl(data[i], data[i + 1], data[i + 2]);
}

// source/core/camera.h:19-34
Camera::Camera(unsigned int width, unsigned int height, float fovY) : pixels{ width, height }
{
    const float o[3] = { 0, 0, 0 }, t[3] = { 0, 0, -1 }, u[3] = { 0, 1, 0 };
    mcrt_camera_look_at(width, height, fovY, o, t, u, &view);
    fov = fovY;
}
void Camera::SetView(vec3 position, vec3 lookAtPosition, vec3 cameraUp)
{
    const float p[3] = { position.x, position.y, position.z }, t[3] = { lookAtPosition.x, lookAtPosition.y, lookAtPosition.z };
    const float u[3] = { cameraUp.x, cameraUp.y, cameraUp.z };
    mcrt_camera_look_at(view.width, view.height, fov, p, t, u, &view);
}

Ray Camera::GetPixelRay(float x, float y) const
{
    // normalised coordinates, y = 0 is the top row; the products are evaluated in double and narrowed (camera.h:40-44)
    vec3 direction(float(-1.0f + x * pixels.deltaX()), float(1.0f - y * pixels.deltaY()), -1.0f);
    direction.x *= view.fov_scale * float(pixels.aspectRatio());        // camera.h:48-49
    direction.y *= view.fov_scale;
    // vec3(viewMatrix * vec4(direction, 0)): glm's mat4 * vec4, include/glm/detail/type_mat4x4.inl (Mul0 + Mul1) + (Mul2 + Mul3)
    const float* m = view.view;
    float r[3];
    for (int k = 0; k < 3; ++k) {
        const float add0 = m[0 + k] * direction.x + m[4 + k] * direction.y;
        const float add1 = m[8 + k] * direction.z + m[12 + k] * 0.0f;
        r[k] = add0 + add1;
    }
    return Ray(vec3(view.position[0], view.position[1], view.position[2]), normalize(vec3(r[0], r[1], r[2])));
}

void Octree::PrintDebug() const
{
    std::printf("octree: %u nodes, %u primitive references, depth %u, %s", nodeCount, primitiveRefs, depth, ldsResident ? "LDS resident" : "read through L2");
    if (buildOnDevice) std::printf(", built on the device in %.2f ms", lastBuildMs);
    std::printf("\r\n");
}

// ---------------------------------------------------------------------- scene
Scene::~Scene()
{
    for (Object* o : objects) delete o;       // lights are aliases, source/scene.cpp:32-36
    if (ctx) mcrt_destroy(ctx);
}

bool Scene::Fail(const char* what) const
{
    std::fprintf(stderr, "mcrt: %s: %s\n", what, mcrt_last_error(ctx));
    return false;
}

bool Scene::EnsureContext()
{
    if (ctx) return true;
    if (mcrt_create(device, &ctx) != MCRT_OK) {
        ctx = nullptr;
        return Fail("no GPU context (this backend has no CPU path)");
    }
    return true;
}

bool Scene::Flatten(mcrt_scene_desc& out)
{
    // light cache, source/scene.cpp:41-49
    lights.clear();
    for (Object* o : objects) {
        const Material& m = o->material;
        if (m.emission.r > 0.0 || m.emission.g > 0.0 || m.emission.b > 0.0) lights.push_back(o);
    }
    for (Object* o : objects) o->UpdateAABB();      // source/scene.cpp:52-55

    flatObjects.clear();
    flatTriangles.clear();
    octree.aabb.Reset();
    for (Object* o : objects) {
        octree.aabb.Encapsulate(o->aabb);           // Octree::Fill root box, octree.h:83-86
        mcrt_object d;
        std::memset(&d, 0, sizeof d);
        d.kind = o->Kind();
        d.position[0] = o->position.x; d.position[1] = o->position.y; d.position[2] = o->position.z;
        d.aabb_min[0] = o->aabb.min.x; d.aabb_min[1] = o->aabb.min.y; d.aabb_min[2] = o->aabb.min.z;
        d.aabb_max[0] = o->aabb.max.x; d.aabb_max[1] = o->aabb.max.y; d.aabb_max[2] = o->aabb.max.z;
        const Material& m = o->material;
        d.material.color[0] = m.color.r; d.material.color[1] = m.color.g; d.material.color[2] = m.color.b;
        d.material.emission[0] = m.emission.r; d.material.emission[1] = m.emission.g; d.material.emission[2] = m.emission.b;
        d.material.surface = int32_t(m.type);
        d.material.diffuse = int32_t(m.diffuse);
        d.material.albedo = m.albedo; d.material.roughness = m.roughness; d.material.refractive_index = m.refractiveIndex;
        if (SphereObject* s = dynamic_cast<SphereObject*>(o)) d.radius = s->radius;
        if (TriangleMesh* mesh = dynamic_cast<TriangleMesh*>(o)) {
            d.first_triangle = uint32_t(flatTriangles.size());
            d.triangle_count = uint32_t(mesh->triangles.size());
            for (const Triangle& t : mesh->triangles) {
                mcrt_triangle ft;
                const vec3* src[4] = { &t.vertex0, &t.vertex1, &t.vertex2, &t.normal };
                float* dst[4] = { ft.v0, ft.v1, ft.v2, ft.normal };
                for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
                flatTriangles.push_back(ft);
            }
            if (Light* l = dynamic_cast<Light*>(o)) {
                d.light_x[0] = l->xVector.x; d.light_x[1] = l->xVector.y; d.light_x[2] = l->xVector.z;
                d.light_y[0] = l->yVector.x; d.light_y[1] = l->yVector.y; d.light_y[2] = l->yVector.z;
            }
        }
        flatObjects.push_back(d);
    }
    out.object_count = uint32_t(flatObjects.size());
    out.triangle_count = uint32_t(flatTriangles.size());
    out.objects = flatObjects.data();
    out.triangles = flatTriangles.data();
    out.background[0] = backgroundColor.r; out.background[1] = backgroundColor.g; out.background[2] = backgroundColor.b;
    return true;
}

void Scene::PrepareForRayTracing()
{
    prepared = false;
    mcrt_scene_desc desc;
    Flatten(desc);
    if (!EnsureContext()) return;
    mcrt_accel_params accel = { octree.maxCount, octree.maxDepth, 0.0f };
    if ((octree.buildOnDevice ? mcrt_upload_scene_device_build(ctx, &desc, &accel) : mcrt_upload_scene(ctx, &desc, &accel)) != MCRT_OK) { Fail("upload_scene"); return; }
    if (octree.buildOnDevice) mcrt_last_build_ms(ctx, &octree.lastBuildMs);
    mcrt_octree_info info;
    if (mcrt_octree_info_get(ctx, &info) == MCRT_OK) {
        octree.nodeCount = info.node_count; octree.primitiveRefs = info.prim_ref_count;
        octree.depth = info.max_depth; octree.ldsResident = info.lds_resident != 0;
    }
    boundCamera = nullptr;
    prepared = true;
}

bool Scene::ObjectsMoved(const std::vector<unsigned int>& indices, bool* rebuilt)
{
    if (!prepared) return Fail("ObjectsMoved before PrepareForRayTracing");
    mcrt_scene_desc desc;
    Flatten(desc);
    std::vector<uint32_t> moved(indices.begin(), indices.end());
    int again = 0;
    if (mcrt_refit_scene(ctx, &desc, (uint32_t)moved.size(), moved.data(), &again) != MCRT_OK) { prepared = false; return Fail("refit_scene"); }
    if (rebuilt) *rebuilt = again != 0;
    mcrt_last_build_ms(ctx, &octree.lastBuildMs);
    mcrt_octree_info info;
    if (mcrt_octree_info_get(ctx, &info) == MCRT_OK) {
        octree.nodeCount = info.node_count; octree.primitiveRefs = info.prim_ref_count;
        octree.depth = info.max_depth; octree.ldsResident = info.lds_resident != 0;
    }
    return true;
}

bool Scene::Bind(Camera& camera)
{
    if (boundCamera == &camera) return true;
    if (mcrt_set_camera(ctx, &camera.view) != MCRT_OK) return Fail("set_camera");
    boundCamera = &camera;
    camera.pixels.rays = 0;
    return true;
}

bool Scene::TraceSamples(Camera& camera, const std::vector<uint32_t>& pixels, const std::vector<uint32_t>& samples, uint64_t seed,
                         unsigned int traceDepth, std::vector<double>& radiance)
{
    if (!prepared) return Fail("TraceSamples before PrepareForRayTracing");
    if (pixels.size() != samples.size()) return Fail("TraceSamples: pixel and sample lists differ in length");
    if (!Bind(camera)) return false;
    radiance.resize(pixels.size() * 3);
    return mcrt_trace_samples(ctx, pixels.size(), pixels.data(), samples.data(), seed, traceDepth, radiance.data()) == MCRT_OK || Fail("trace_samples");
}

bool Scene::ResolvePerPixelCounts(Camera& camera, int tonemap, double gamma, double exposure, std::vector<uint8_t>& bgra8)
{
    if (!prepared) return Fail("Resolve before PrepareForRayTracing");
    if (!Bind(camera)) return false;
    // the device resolves sums / spp_total with one count for the image: hand it the per-pixel means and a count of 1
    // (x / 1.0 is exact), so the tonemap and the 8-bit store stay the device's
    PixelBuffer& pb = camera.pixels;
    std::vector<double> mean(pb.data.size(), 0.0);
    for (size_t p = 0; p < size_t(pb.numPixels()); ++p) {
        const uint64_t n = pb.rays + (pb.rayCount.empty() ? 0u : pb.rayCount[p]);
        if (n == 0) continue;
        for (size_t k = 0; k < 3; ++k) mean[p * 3 + k] = pb.data[p * 3 + k] / double(n);
    }
    // the device accumulator is borrowed for the means and put back afterwards
    std::vector<double> keep(pb.data.size());
    if (mcrt_read_accumulator(ctx, keep.data()) != MCRT_OK) return Fail("read_accumulator");
    if (mcrt_write_accumulator(ctx, mean.data()) != MCRT_OK) return Fail("write_accumulator");
    bgra8.resize(size_t(pb.numPixels()) * 4);
    const bool ok = mcrt_resolve(ctx, tonemap, gamma, exposure, 1, bgra8.data(), nullptr) == MCRT_OK;
    if (mcrt_write_accumulator(ctx, keep.data()) != MCRT_OK) return Fail("write_accumulator");
    return ok || Fail("resolve");
}

bool Scene::Reduce(Camera& camera, void* ncclComm, int root)
{
    if (boundCamera != &camera) return Fail("Reduce: camera not rendered with this scene");
    return mcrt_reduce(ctx, ncclComm, root) == MCRT_OK || Fail("reduce");
}

bool Scene::AddAccumulatorOf(Camera& camera, Scene& other)
{
    if (boundCamera != &camera || !other.ctx) return Fail("AddAccumulatorOf: camera not rendered with this scene");
    if (other.device != device) return Fail("AddAccumulatorOf: the other context lives on another GPU (use Reduce)");
    double* src = nullptr;
    if (mcrt_synchronize(other.ctx) != MCRT_OK || mcrt_accumulator(other.ctx, &src) != MCRT_OK) return Fail("AddAccumulatorOf: other context");
    return mcrt_add_accumulator(ctx, src) == MCRT_OK || Fail("add_accumulator");
}

bool Scene::IntersectRay(Ray& ray, RayIntersectionInfo& hitInfo) const
{
    hitInfo.Reset();
    if (!prepared) return Fail("IntersectRay before PrepareForRayTracing");
    int32_t obj = -1; uint32_t elem = 0; float t = FLOAT_INFINITY;
    if (mcrt_intersect(ctx, 1, &ray.origin.x, &ray.direction.x, &obj, &elem, &t) != MCRT_OK) return Fail("intersect");
    if (obj < 0) return false;
    hitInfo.object = objects[size_t(obj)];
    hitInfo.elementIndex = elem;
    hitInfo.hitDistance = t;
    return true;
}

// ---- the per-object virtuals, source/objects/object.h:23-29, forwarded to the device
bool Scene::ObjectIntersects(unsigned int index, vec3 rayOrigin, vec3 rayDirection, unsigned int& elementIndex, float& hitDistance) const
{
    if (!prepared) return Fail("Object::Intersects before PrepareForRayTracing");
    uint8_t hit = 0; uint32_t elem = 0; float t = FLOAT_INFINITY;
    if (mcrt_object_intersect(ctx, index, 1, &rayOrigin.x, &rayDirection.x, &hit, &elem, &t) != MCRT_OK) return Fail("object_intersect");
    elementIndex = elem; hitDistance = t;
    return hit != 0;
}

bool Scene::ObjectSurfaceNormal(unsigned int index, vec3 location, unsigned int element, vec3& normal) const
{
    if (!prepared) return Fail("Object::GetSurfaceNormal before PrepareForRayTracing");
    const uint32_t ob = index, el = element;
    return mcrt_surface_normals(ctx, 1, &ob, &el, &location.x, &normal.x) == MCRT_OK || Fail("surface_normals");
}

bool Scene::ObjectRandomPoint(unsigned int index, UniformRandomGenerator& gen, vec3& point) const
{
    if (!prepared) return Fail("Object::GetRandomPointOnSurface before PrepareForRayTracing");
    return mcrt_random_surface_points(ctx, index, 1, gen.xorseed, &point.x) == MCRT_OK || Fail("random_surface_points");
}

static bool Orphan(const char* what)
{
    std::fprintf(stderr, "mcrt: %s: the object was not created by Scene::CreateObject\n", what);
    return false;
}

bool Object::Intersects(vec3 rayOrigin, vec3 rayDirection, RayIntersectionInfo& hitInfo)
{
    unsigned int element = 0;
    float distance = FLOAT_INFINITY;
    const bool hit = scene ? scene->ObjectIntersects(sceneIndex, rayOrigin, rayDirection, element, distance) : Orphan("Object::Intersects");
    if (hit) { hitInfo.object = this; hitInfo.elementIndex = element; hitInfo.hitDistance = distance; }
    // a missed sphere leaves hitInfo as it was (source/objects/sphere.h:18-45), a missed mesh resets it (source/objects/mesh.cpp:14,36)
    else if (Kind() != MCRT_OBJECT_SPHERE) hitInfo.Reset();
    return hit;
}

vec3 Object::GetSurfaceNormal(vec3 location, unsigned int index)
{
    vec3 n{ 0.0f };
    if (scene) scene->ObjectSurfaceNormal(sceneIndex, location, index, n); else Orphan("Object::GetSurfaceNormal");
    return n;
}

vec3 Object::GetRandomPointOnSurface(UniformRandomGenerator& gen)
{
    vec3 p = position;
    if (scene) scene->ObjectRandomPoint(sceneIndex, gen, p); else Orphan("Object::GetRandomPointOnSurface");
    return p;
}

// ---- Material::BRDF, source/core/material.h:24-53
namespace {
int g_materialDevice = 0;
mcrt_ctx* MaterialContext()
{
    static mcrt_ctx* ctx = nullptr;
    static int ctxDevice = -1;
    if (ctx && ctxDevice != g_materialDevice) { mcrt_destroy(ctx); ctx = nullptr; }
    if (!ctx) {
        if (mcrt_create(g_materialDevice, &ctx) != MCRT_OK) {
            std::fprintf(stderr, "mcrt: Material::BRDF: %s\n", mcrt_last_error(nullptr));
            ctx = nullptr;
        }
        ctxDevice = g_materialDevice;
    }
    return ctx;
}
} // namespace

void MaterialDevice(int deviceId) { g_materialDevice = deviceId; }

double Material::BRDF(vec3& incident, vec3& reflection, vec3& normal)
{
    mcrt_ctx* c = MaterialContext();
    if (!c) return 0.0;
    mcrt_material m;
    std::memset(&m, 0, sizeof m);
    m.diffuse = int32_t(diffuse); m.surface = int32_t(type);
    m.albedo = albedo; m.roughness = roughness; m.refractive_index = refractiveIndex;
    double out = 0.0;
    if (mcrt_brdf(c, &m, 1, &incident.x, &reflection.x, &normal.x, &out) != MCRT_OK) {
        std::fprintf(stderr, "mcrt: Material::BRDF: %s\n", mcrt_last_error(c));
        return 0.0;
    }
    return out;
}

// source/scene.cpp:82-92
ColorDbl Scene::TraceUnlit(Ray ray) const
{
    RayIntersectionInfo hitInfo;
    if (IntersectRay(ray, hitInfo)) return hitInfo.object->material.color;
    return ColorDbl{ 0.0 };
}

ColorDbl Scene::TraceRay(Ray ray, UniformRandomGenerator& uniformGenerator, unsigned int traceDepth, ColorDbl importance)
{
    if (!prepared) { Fail("TraceRay before PrepareForRayTracing"); return ColorDbl{ 0.0 }; }
    double out[3] = { 0, 0, 0 };
    if (mcrt_trace_rays(ctx, 1, &ray.origin.x, &ray.direction.x, uniformGenerator.xorseed, traceDepth, &importance.r, out) != MCRT_OK) {
        Fail("trace_rays");
        return ColorDbl{ 0.0 };
    }
    return ColorDbl(out[0], out[1], out[2]);
}

bool Scene::Render(Camera& camera, uint32_t sampleBegin, uint32_t sampleCount, unsigned int traceDepth, uint64_t seed)
{
    if (!prepared) return Fail("Render before PrepareForRayTracing");
    if (!Bind(camera)) return false;
    if (mcrt_render(ctx, sampleBegin, sampleCount, traceDepth, seed) != MCRT_OK) return Fail("render");
    camera.pixels.rays += sampleCount;
    return true;
}

bool Scene::Clear(Camera& camera)
{
    if (boundCamera != &camera) return true;
    camera.pixels.rays = 0;
    return mcrt_clear(ctx) == MCRT_OK || Fail("clear");
}

bool Scene::ReadBack(Camera& camera)
{
    if (boundCamera != &camera) return Fail("ReadBack: camera not rendered with this scene");
    return mcrt_read_accumulator(ctx, camera.pixels.data.data()) == MCRT_OK || Fail("read_accumulator");
}

bool Scene::Restore(Camera& camera, const std::vector<double>& sums, uint64_t samplesDone)
{
    if (!prepared) return Fail("Restore before PrepareForRayTracing");
    if (sums.size() != size_t(camera.pixels.numPixels()) * 3) return Fail("Restore: wrong accumulator size");
    if (!Bind(camera)) return false;
    if (mcrt_write_accumulator(ctx, sums.data()) != MCRT_OK) return Fail("write_accumulator");
    camera.pixels.rays = samplesDone;
    return true;
}

bool Scene::Resolve(Camera& camera, int tonemap, double gamma, double exposure, std::vector<uint8_t>& bgra8)
{
    if (boundCamera != &camera || camera.pixels.rays == 0) return Fail("Resolve: nothing rendered with this camera");
    bgra8.resize(size_t(camera.pixels.numPixels()) * 4);
    return mcrt_resolve(ctx, tonemap, gamma, exposure, camera.pixels.rays, bgra8.data(), nullptr) == MCRT_OK || Fail("resolve");
}

// source/scene.cpp:236-239
void Scene::MoveCameraToRecommendedPosition(Camera& camera)
{
    camera.SetView(vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 0.0f, 1.0f), vec3(0.0f, 1.0f, 0.0f));
}

// ----------------------------------------------------------- example scenes
// HexagonScene, source/scene.cpp:242-366
HexagonScene::HexagonScene()
{
    ceiling = CreateObject<TriangleMesh>();
    floor = CreateObject<TriangleMesh>();
    walls1 = CreateObject<TriangleMesh>();
    walls2 = CreateObject<TriangleMesh>();
    walls3 = CreateObject<TriangleMesh>();
    ceiling->material.color = ColorDbl(0.2);
    floor->material.color = ColorDbl(0.2);
    walls1->material.color = ColorDbl(0.2, 0.01, 0.01);
    walls2->material.color = ColorDbl(0.01, 0.2, 0.01);
    walls3->material.color = ColorDbl(0.2);

    // six ceiling corners (x, y, z); the floor mirrors them in y
    const float cx[6] = { 0.0f, 6.0f, 6.0f, 0.0f, -6.0f, -6.0f };
    const float cz[6] = { -3.0f, 0.0f, 10.0f, 13.0f, 10.0f, 0.0f };
    vec3 c[6], f[6];
    for (int i = 0; i < 6; ++i) { c[i] = vec3(cx[i], 5.0f, cz[i]); f[i] = c[i]; f[i].y *= -1.0f; }

    floor->AddQuad(f[0], f[3], f[2], f[1]);
    floor->AddQuad(f[0], f[5], f[4], f[3]);
    ceiling->AddQuad(c[0], c[1], c[2], c[3]);
    ceiling->AddQuad(c[3], c[4], c[5], c[0]);
    walls1->AddQuad(f[1], f[2], c[2], c[1]);
    walls2->AddQuad(f[0], f[1], c[1], c[0]);
    walls3->AddQuad(f[2], f[3], c[3], c[2]);
    walls2->AddQuad(f[4], f[5], c[5], c[4]);
    walls3->AddQuad(f[3], f[4], c[4], c[3]);
    walls3->AddQuad(f[5], f[0], c[0], c[5]);
}

void HexagonScene::MoveCameraToRecommendedPosition(Camera& camera)
{
    camera.SetView(vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 0.0f, 10.0f), vec3(0.0f, 1.0f, 0.0f));
}

void HexagonScene::AddExampleObjects(float radius)
{
    SphereObject* spheres[3];
    const SurfaceType kinds[3] = { SurfaceType::Diffuse, SurfaceType::Specular, SurfaceType::Refractive };
    const vec3 where[3] = { vec3(3.0, 2.0, 10.0f), vec3(-3.0f, 0.0, 8.0f), vec3(1.0f, -3.0, 6.0f) };
    for (int i = 0; i < 3; ++i) {
        spheres[i] = CreateObject<SphereObject>();
        spheres[i]->material.type = kinds[i];
        spheres[i]->radius = radius;
        spheres[i]->position = where[i];
        spheres[i]->material.color = ColorDbl(0.5);
    }
    Box* box = CreateObject<Box>();
    box->SetGeometry({ 3.0, -5.0, 10.0 }, { 0.0f, 1.0f, 0.0f }, { 0.5f, 0.0f, 1.0f }, 2.0f, 2.0f, 7.0f - radius);
    box->material.color = ColorDbl(0.01, 0.3, 0.8);
    box->material.type = SurfaceType::Diffuse;
    Box* box2 = CreateObject<Box>();
    box2->SetGeometry({ -3.0, -5.0, 8.0 }, { 0.0f, 1.0f, 0.0f }, { 0.5f, 0.0f, 1.0f }, 2.0f, 2.0f, 5.0f - radius);
    box2->material.color = ColorDbl(0.8, 0.4, 0.01);
    box2->material.type = SurfaceType::Refractive;
    Box* box3 = CreateObject<Box>();
    box3->SetGeometry({ 1.0, -5.0, 6.0 }, { 0.0f, 1.0f, 0.0f }, { 0.5f, 0.0f, 1.0f }, 4.0f, 4.0f, 2.0f - radius);
    box3->material.color = ColorDbl(0.5, 0.2, 0.8);
    box3->material.type = SurfaceType::Diffuse;
    spheres[2]->material.refractiveIndex = 1.52f;
    box2->material.refractiveIndex = 1.52f;
}

namespace {
// the two light variants shared by both example scenes, source/scene.cpp:344-366,481-503
void AddLightAt(Scene& scene, vec3 roofCenter, vec2 dims, ColorDbl lightColor, bool usePoint)
{
    if (usePoint) {
        SphereObject* pointLight = scene.CreateObject<SphereObject>();
        pointLight->radius = 0.0f;
        pointLight->material.color = lightColor;
        pointLight->material.emission = lightColor;
        pointLight->position = roofCenter;
    } else {
        Light* light = scene.CreateObject<Light>();
        light->SetGeometry(roofCenter, { 0.0f, -1.0f, 0.0f }, { 1.0f, 0.0f, 0.0f }, dims);
        light->material.color = lightColor;
        light->material.emission = lightColor;
    }
}
} // namespace

void HexagonScene::AddExampleLight(ColorDbl lightColor, bool usePoint)
{
    AddLightAt(*this, vec3{ 0.0f, 5.0f - 0.001f, 8.0f }, vec2{ 1.0f, 1.0f }, lightColor, usePoint);
}

// CornellBoxScene, source/scene.cpp:370-503.  The room as data: eight corners (index = 4 * floor + corner, corners 0..3 run
// round the ceiling from (-w, +h, +l)) and five quads over them; object order - left wall, right wall, the white mesh - is the
// reference's (it is the tie-break order of the nearest hit).
CornellBoxScene::CornellBoxScene(float length, float width, float height)
{
    halfLength = length / 2.0f;
    halfWidth = width / 2.0f;
    halfHeight = height / 2.0f;
    vec3 corner[8];
    const float sx[4] = { -1.0f, 1.0f, 1.0f, -1.0f }, sz[4] = { 1.0f, 1.0f, -1.0f, -1.0f };
    for (int k = 0; k < 4; ++k) {
        corner[k] = vec3(sx[k] * halfWidth, halfHeight, sz[k] * halfLength);
        corner[4 + k] = corner[k];
        corner[4 + k].y *= -1.0f;                     // the floor corners: the ceiling's with the height flipped (scene.cpp:395-398)
    }
    struct Wall { TriangleMesh** mesh; ColorDbl color; };
    const Wall walls[3] = { { &leftWall, ColorDbl(0.2, 0.01, 0.01) }, { &rightWall, ColorDbl(0.01, 0.2, 0.01) }, { &whiteSegments, ColorDbl(0.2) } };
    for (const Wall& w : walls) {
        *w.mesh = CreateObject<TriangleMesh>();
        (*w.mesh)->material.color = w.color;
    }
    struct Quad { int wall, a, b, c, d; };
    const Quad quads[5] = { { 0, 5, 1, 2, 6 }, { 1, 4, 7, 3, 0 }, { 2, 3, 2, 1, 0 } /* ceiling */, { 2, 7, 6, 2, 3 } /* back wall */, { 2, 4, 5, 6, 7 } /* floor */ };
    for (const Quad& q : quads) (*walls[q.wall].mesh)->AddQuad(corner[q.a], corner[q.b], corner[q.c], corner[q.d]);
}

void CornellBoxScene::MoveCameraToRecommendedPosition(Camera& camera)
{
    camera.SetView(vec3(0.0f, 0.0f, halfLength), vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f));
}

// The example objects as two tables (source/scene.cpp:413-472): four spheres, then three boxes, in the reference's creation
// order; every coordinate expression is the reference's own (mixed float / double arithmetic included), evaluated once.
void CornellBoxScene::AddExampleObjects(float radius)
{
    const float widthOffset = halfWidth - radius;
    const float depthOffset = halfLength - radius;
    struct SphereSpec { vec3 position; SurfaceType type; DiffuseType diffuse; float roughness, ior; };
    const Material plain;
    const SphereSpec spheres[4] = {
        { vec3(-widthOffset, 0.0f, -depthOffset / 2.0), SurfaceType::Diffuse, DiffuseType::Lambertian, plain.roughness, plain.refractiveIndex },
        { vec3(0, 2.0, -halfLength + radius), SurfaceType::Specular, DiffuseType::Lambertian, plain.roughness, plain.refractiveIndex },
        { vec3(widthOffset, 0.0f, -depthOffset / 2.0), SurfaceType::Diffuse, DiffuseType::OrenNayar, 0.5f, plain.refractiveIndex },
        { vec3(0, -halfHeight + radius + 1.5, -2), SurfaceType::Refractive, DiffuseType::Lambertian, plain.roughness, 1.4f },
    };
    for (const SphereSpec& spec : spheres) {
        SphereObject* s = CreateObject<SphereObject>();
        s->radius = radius;
        s->position = spec.position;
        s->material.color = ColorDbl(0.5);
        s->material.type = spec.type;
        s->material.diffuse = spec.diffuse;
        s->material.roughness = spec.roughness;
        s->material.refractiveIndex = spec.ior;
    }
    struct BoxSpec { vec3 base, side; float width, depth, height; ColorDbl color; DiffuseType diffuse; float roughness; };
    const BoxSpec boxes[3] = {
        { { halfWidth - 1.5f, -halfHeight, -depthOffset / 2.0 }, { -0.5f, 0.0f, 1.0f }, 2.0f, 2.0f, halfHeight - radius, ColorDbl(0.01, 0.3, 0.8), DiffuseType::OrenNayar, 0.5f },
        { { -halfWidth + 1.5f, -halfHeight, -depthOffset / 2.0 }, { 0.5f, 0.0f, 1.0f }, 2.0f, 2.0f, halfHeight - radius, ColorDbl(0.8, 0.4, 0.01), DiffuseType::Lambertian, plain.roughness },
        { { 0, -halfHeight, -3 }, { 1.0f, 0.0f, 1.0f }, 4.0f, 4.0f, halfHeight - radius - 2.3f, ColorDbl(0.5, 0.2, 0.8), DiffuseType::Lambertian, plain.roughness },
    };
    for (const BoxSpec& spec : boxes) {
        Box* box = CreateObject<Box>();
        box->SetGeometry(spec.base, { 0.0f, 1.0f, 0.0f }, spec.side, spec.width, spec.depth, spec.height);
        box->material.color = spec.color;
        box->material.diffuse = spec.diffuse;
        box->material.roughness = spec.roughness;
    }
}

void CornellBoxScene::AddExampleLight(ColorDbl lightColor, bool usePoint)
{
    AddLightAt(*this, vec3{ 0.0f, halfHeight - 0.001f, 0.0f }, vec2{ halfWidth / 3.0f, halfHeight / 3.0f }, lightColor, usePoint);
}

// ------------------------------------------------------------------- screenshot (source/opengl/screenshot.cpp)
namespace {
uint32_t Crc32(uint32_t crc, const uint8_t* p, size_t n)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void PutBE32(std::vector<uint8_t>& out, uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back(uint8_t(v >> s)); }
void PutChunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data)
{
    PutBE32(out, uint32_t(data.size()));
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    PutBE32(out, Crc32(0u, out.data() + at, out.size() - at));
}
} // namespace

bool WritePNG(const std::string& path, const std::vector<uint8_t>& bgra8, unsigned width, unsigned height)
{
    if (width == 0 || height == 0 || bgra8.size() < size_t(width) * height * 4) return false;
    // scanlines: filter byte 0 + RGBA
    std::vector<uint8_t> raw;
    raw.reserve((size_t(width) * 4 + 1) * height);
    for (unsigned y = 0; y < height; ++y) {
        raw.push_back(0);
        for (unsigned x = 0; x < width; ++x) {
            const uint8_t* p = &bgra8[(size_t(y) * width + x) * 4];
            raw.push_back(p[2]); raw.push_back(p[1]); raw.push_back(p[0]); raw.push_back(p[3]);
        }
    }
    // zlib stream of stored blocks (at most 65535 bytes each) + Adler-32
    std::vector<uint8_t> z = { 0x78, 0x01 };
    uint32_t a = 1, b = 0;
    for (size_t at = 0; at < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8)); z.push_back(uint8_t(~n)); z.push_back(uint8_t(~n >> 8));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        for (size_t i = at; i < at + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        at += n;
    }
    PutBE32(z, (b << 16) | a);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    std::vector<uint8_t> ihdr;
    PutBE32(ihdr, width); PutBE32(ihdr, height);
    const uint8_t tail[5] = { 8, 6, 0, 0, 0 };       // 8 bits per channel, RGBA, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    PutChunk(out, "IHDR", ihdr);
    PutChunk(out, "IDAT", z);
    PutChunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0) && ok;
}

std::string TakeScreenshot(std::string filename, const std::vector<uint8_t>& bgra8, unsigned screenWidth, unsigned screenHeight)
{
    const size_t lastdot = filename.find_last_of(".");
    if (lastdot != std::string::npos) filename = filename.substr(0, lastdot);      // screenshot.cpp:18-24
    const std::string baseName = filename;
    int count = 0;
    do {                                                                           // screenshot.cpp:31-37
        count++;
        const std::string digits = std::to_string(count);
        filename = baseName + std::string(digits.length() < 5 ? 5 - digits.length() : 0, '0') + digits + ".png";
    } while (std::ifstream(filename).good());
    if (!WritePNG(filename, bgra8, screenWidth, screenHeight)) {
        std::cout << "encoder error: cannot write " << filename << std::endl;
        return std::string();
    }
    std::cout << "Screenshot saved to " + filename + "\r\n";
    return filename;
}

// ------------------------------------------------------------------- C shim
// Lets tests (ctypes) build the example scenes through this layer and compare the flat description
// with dumps of the reference's own objects.
extern "C" {

void* mcrt_host_scene_create(int which, float radius, double light_strength, int use_point_light)
{
    if (which == 0) {
        CornellBoxScene* s = new CornellBoxScene(10.0f, 10.0f, 10.0f);   // source/main.cpp:237-241
        s->backgroundColor = { 0.0, 0.0, 0.0 };
        s->AddExampleObjects(radius);
        s->AddExampleLight(ColorDbl{ light_strength }, use_point_light != 0);
        return static_cast<Scene*>(s);
    }
    HexagonScene* s = new HexagonScene();
    s->AddExampleObjects(radius);
    s->AddExampleLight(ColorDbl{ light_strength }, use_point_light != 0);
    return static_cast<Scene*>(s);
}

void mcrt_host_scene_destroy(void* scene) { delete static_cast<Scene*>(scene); }

// Object added through the public API: a TriangleMesh loaded from an OBJ file.
int mcrt_host_scene_add_obj(void* scene, const char* path, float px, float py, float pz, double grey)
{
    TriangleMesh* m = static_cast<Scene*>(scene)->CreateObject<TriangleMesh>();
    m->position = vec3(px, py, pz);
    m->material.color = ColorDbl(grey);
    return m->LoadMesh(path) ? int(m->triangles.size()) : -1;
}

// Pointers stay valid until the next call on the same scene.
// Camera::Camera + SetView + GetPixelRay on the host, for n pixel positions (test hook)
void mcrt_host_camera_rays(unsigned width, unsigned height, float fov_y, const float pos[3], const float target[3], const float up[3], uint64_t n,
                           const float* xs, const float* ys, float* out_o, float* out_d)
{
    Camera cam{ width, height, fov_y };
    cam.SetView(vec3(pos[0], pos[1], pos[2]), vec3(target[0], target[1], target[2]), vec3(up[0], up[1], up[2]));
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = cam.GetPixelRay(xs[i], ys[i]);
        out_o[3 * i] = r.origin.x; out_o[3 * i + 1] = r.origin.y; out_o[3 * i + 2] = r.origin.z;
        out_d[3 * i] = r.direction.x; out_d[3 * i + 1] = r.direction.y; out_d[3 * i + 2] = r.direction.z;
    }
}

// UniformRandomGenerator::RandomDouble on the host (test hook)
void mcrt_host_rng_doubles(uint64_t s0, uint64_t s1, uint64_t n, double* out)
{
    UniformRandomGenerator g(s0, s1);
    for (uint64_t i = 0; i < n; ++i) out[i] = g.RandomDouble();
}

// PixelBuffer::Accumulate / GetPixelColor / GetRayCount (test hook): n samples added one by one
void mcrt_host_accumulate(unsigned width, unsigned height, uint64_t n, const uint32_t* xs, const uint32_t* ys, const double* colors, double* out_sums,
                          uint64_t* out_counts)
{
    PixelBuffer pb{ width, height };
    for (uint64_t i = 0; i < n; ++i) pb.Accumulate(xs[i], ys[i], ColorDbl(colors[3 * i], colors[3 * i + 1], colors[3 * i + 2]));
    for (unsigned y = 0; y < height; ++y)
        for (unsigned x = 0; x < width; ++x) {
            const ColorDbl c = pb.GetPixelColor(x, y);
            double* o = out_sums + (size_t(y) * width + x) * 3;
            o[0] = c.r; o[1] = c.g; o[2] = c.b;
            out_counts[size_t(y) * width + x] = pb.GetRayCount(x, y);
        }
}

// TakeScreenshot (test hook): the name written is copied to out_name (empty string on failure)
int mcrt_host_take_screenshot(const char* filename, const uint8_t* bgra8, unsigned width, unsigned height, char* out_name, unsigned out_cap)
{
    const std::vector<uint8_t> image(bgra8, bgra8 + size_t(width) * height * 4);
    const std::string name = TakeScreenshot(filename, image, width, height);
    if (out_name && out_cap) { std::strncpy(out_name, name.c_str(), out_cap - 1); out_name[out_cap - 1] = 0; }
    return name.empty() ? 1 : 0;
}

// Object::Intersects and GetSurfaceNormal through the virtual interface, one call per ray (test hook): object `index` of the scene
int mcrt_host_object_intersects(void* scene, unsigned index, uint64_t n, const float* o, const float* d, uint8_t* out_hit, uint32_t* out_elem, float* out_t,
                                float* out_normal)
{
    Scene* s = static_cast<Scene*>(scene);
    if (index >= s->ObjectCount()) return 1;
    Object* ob = s->GetObject(index);
    for (uint64_t i = 0; i < n; ++i) {
        RayIntersectionInfo info;
        const vec3 ro(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        const bool hit = ob->Intersects(ro, rd, info);
        if (hit != (info.object == ob)) return 2;
        out_hit[i] = hit ? 1 : 0; out_elem[i] = info.elementIndex; out_t[i] = info.hitDistance;
        vec3 nrm{ 0.0f };
        if (hit) nrm = ob->GetSurfaceNormal(ro + rd * info.hitDistance, info.elementIndex);
        out_normal[3 * i] = nrm.x; out_normal[3 * i + 1] = nrm.y; out_normal[3 * i + 2] = nrm.z;
    }
    return 0;
}

// Object::GetRandomPointOnSurface through the virtual interface (test hook): one generator, n consecutive points; the final state is returned
int mcrt_host_object_random_points(void* scene, unsigned index, uint64_t n, uint64_t state[2], float* out)
{
    Scene* s = static_cast<Scene*>(scene);
    if (index >= s->ObjectCount()) return 1;
    UniformRandomGenerator g(state[0], state[1]);
    for (uint64_t i = 0; i < n; ++i) {
        const vec3 p = s->GetObject(index)->GetRandomPointOnSurface(g);
        out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
    }
    state[0] = g.xorseed[0]; state[1] = g.xorseed[1];
    return 0;
}

// Material::BRDF (test hook)
void mcrt_host_material_brdf(int diffuse, float albedo, float roughness, uint64_t n, const float* inc, const float* refl, const float* nrm, double* out)
{
    Material m;
    m.diffuse = DiffuseType(diffuse); m.albedo = albedo; m.roughness = roughness;
    for (uint64_t i = 0; i < n; ++i) {
        vec3 a(inc[3 * i], inc[3 * i + 1], inc[3 * i + 2]), b(refl[3 * i], refl[3 * i + 1], refl[3 * i + 2]), c(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
        out[i] = m.BRDF(a, b, c);
    }
}

// const Scene::TraceUnlit (source/scene.h:43) through a const reference (test hook)
void mcrt_host_trace_unlit(void* scene, uint64_t n, const float* o, const float* d, double* out)
{
    const Scene& s = *static_cast<Scene*>(scene);
    for (uint64_t i = 0; i < n; ++i) {
        const ColorDbl c = s.TraceUnlit(Ray(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2])));
        out[3 * i] = c.r; out[3 * i + 1] = c.g; out[3 * i + 2] = c.b;
    }
}

// Scene::ObjectsMoved (test hook): object `index` translated by (dx, dy, dz) - a sphere's centre, a mesh's vertices - then the refit
int mcrt_host_scene_move_object(void* scene, unsigned index, float dx, float dy, float dz, int* out_rebuilt)
{
    Scene* s = static_cast<Scene*>(scene);
    if (index >= s->ObjectCount()) return 1;
    Object* ob = s->GetObject(index);
    const vec3 delta(dx, dy, dz);
    ob->position = ob->position + delta;
    if (TriangleMesh* mesh = dynamic_cast<TriangleMesh*>(ob))
        for (Triangle& t : mesh->triangles) t = Triangle(t.vertex0 + delta, t.vertex1 + delta, t.vertex2 + delta);
    bool rebuilt = false;
    if (!s->ObjectsMoved({ index }, &rebuilt)) return 2;
    if (out_rebuilt) *out_rebuilt = rebuilt ? 1 : 0;
    return 0;
}

int mcrt_host_scene_flatten(void* scene, mcrt_scene_desc* out) { return static_cast<Scene*>(scene)->Flatten(*out) ? 0 : 1; }

int mcrt_host_scene_prepare(void* scene)
{
    Scene* s = static_cast<Scene*>(scene);
    s->PrepareForRayTracing();
    return s->Context() ? 0 : 1;
}

} // extern "C"
