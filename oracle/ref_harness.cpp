/*
 * ref_harness.cpp - TEST INFRASTRUCTURE, not product code.
 *
 * A headless C-ABI driver around the REAL reference translation units
 * (source/scene.cpp, source/objects/mesh.cpp, source/core/randomization.cpp,
 * source/core/pixelbuffer.cpp), which are compiled where they lie under
 * /root/reference by oracle/Makefile into oracle/_ref/libmcrt_ref.so.  The
 * reference has no headless mode, no seed control and no test hooks
 * (source/main.cpp:31-50, source/core/randomization.cpp:8-13), so this file
 * supplies exactly those and nothing else: every arithmetic result it returns
 * is produced by the reference's own code.
 *
 * Used only by tests/, oracle/gen_golden.py and bench.py's cpu_baseline leg.
 */
#include "scene.h"
#include "objects/sphere.h"
#include "objects/box.h"
#include "objects/light.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "mcrt.h"

namespace {

/* UniformRandomGenerator::xorseed is protected (source/core/randomization.h:15-16). */
struct SeedableRng : UniformRandomGenerator {
    void Seed(uint64_t a, uint64_t b) { xorseed[0] = a; xorseed[1] = b; }
    void State(uint64_t out[2]) const { out[0] = xorseed[0]; out[1] = xorseed[1]; }
};

/* Scene::objects / lights / RandomHemisphereRay are protected (source/scene.h:18-22). */
struct SceneAccess : Scene {
    static std::vector<Object*>& Objects(Scene& s) { return s.*(&SceneAccess::objects); }
    static std::vector<Object*>& Lights(Scene& s) { return s.*(&SceneAccess::lights); }
    static Ray Hemisphere(Scene& s, vec3& o, vec3& in, vec3& n, UniformRandomGenerator& g, float& c)
    {
        return (s.*(&SceneAccess::RandomHemisphereRay))(o, in, n, g, c);
    }
};

/* Camera::viewMatrix / fovPixelScale / position are protected (source/core/camera.h:11-14). */
struct CameraAccess : Camera {
    using Camera::Camera;
    const glm::mat4& View() const { return viewMatrix; }
    float FovScale() const { return fovPixelScale; }
    vec3 Position() const { return position; }
};

struct RefScene {
    Scene* scene = nullptr;
};

/* Same counter -> state mapping as mcrt_rng_seed (include/mcrt.h); kept as an
 * independent copy so the harness links nothing from the product. */
inline uint64_t Mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline void SampleSeed(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t st[2])
{
    uint64_t z = seed + uint64_t(pixel) * 0x9E3779B97F4A7C15ull + uint64_t(sample) * 0xD1B54A32D192ED03ull;
    st[0] = Mix64(z + 0x9E3779B97F4A7C15ull);
    st[1] = Mix64(z + 0x3C6EF372FE94F82Aull);
    if ((st[0] | st[1]) == 0) st[1] = 1;
}

void CopyMaterialOut(const Material& m, mcrt_material& o)
{
    o.color[0] = m.color.r; o.color[1] = m.color.g; o.color[2] = m.color.b;
    o.emission[0] = m.emission.r; o.emission[1] = m.emission.g; o.emission[2] = m.emission.b;
    o.surface = int32_t(m.type);
    o.diffuse = int32_t(m.diffuse);
    o.albedo = m.albedo;
    o.roughness = m.roughness;
    o.refractive_index = m.refractiveIndex;
    o.reserved = 0;
}
void CopyMaterialIn(const mcrt_material& o, Material& m)
{
    m.color = ColorDbl(o.color[0], o.color[1], o.color[2]);
    m.emission = ColorDbl(o.emission[0], o.emission[1], o.emission[2]);
    m.type = SurfaceType(o.surface);
    m.diffuse = DiffuseType(o.diffuse);
    m.albedo = o.albedo;
    m.roughness = o.roughness;
    m.refractiveIndex = o.refractive_index;
}
inline void Put3(float* d, const vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline vec3 Get3(const float* s) { return vec3(s[0], s[1], s[2]); }

bool Finite3(const ColorDbl& c) { return std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z); }

} // namespace

extern "C" {

/* The default scene exactly as source/main.cpp:234-242 builds it. */
void* ref_scene_create_cornell(float length, float width, float height, float sphere_radius,
                               double light_strength, int use_point_light)
{
    CornellBoxScene* s = new CornellBoxScene(length, width, height);
    s->backgroundColor = { 0.0f, 0.0f, 0.0f };
    s->AddExampleObjects(sphere_radius);
    s->AddExampleLight(ColorDbl{ light_strength }, use_point_light != 0);
    s->PrepareForRayTracing();
    RefScene* h = new RefScene;
    h->scene = s;
    return h;
}

void* ref_scene_create_hexagon(float sphere_radius, double light_strength, int use_point_light)
{
    HexagonScene* s = new HexagonScene();
    s->AddExampleObjects(sphere_radius);
    s->AddExampleLight(ColorDbl{ light_strength }, use_point_light != 0);
    s->PrepareForRayTracing();
    RefScene* h = new RefScene;
    h->scene = s;
    return h;
}

/* Rebuild a scene from the flat description through the reference's own
 * classes (Scene::CreateObject, Triangle ctor, PrepareForRayTracing).
 * build_reference_octree == 0 skips Octree::Fill: the reference's octree is
 * never queried (source/scene.cpp:64) and its recursive build
 * (source/accelerationstructures/octree.h:50-70,168-223) overflows the stack
 * on some scenes (e.g. a 2000-triangle mesh object among the walls); the
 * other two steps of PrepareForRayTracing (light cache, UpdateAABB,
 * source/scene.cpp:41-55) are then run here through the same public members. */
void* ref_scene_create_from_desc(const mcrt_scene_desc* d, int build_reference_octree)
{
    Scene* s = new Scene();
    s->backgroundColor = ColorDbl(d->background[0], d->background[1], d->background[2]);
    for (uint32_t i = 0; i < d->object_count; ++i) {
        const mcrt_object& o = d->objects[i];
        if (o.kind == MCRT_OBJECT_SPHERE) {
            SphereObject* sp = s->CreateObject<SphereObject>();
            sp->position = Get3(o.position);
            sp->radius = o.radius;
            CopyMaterialIn(o.material, sp->material);
        } else {
            TriangleMesh* mesh;
            if (o.kind == MCRT_OBJECT_LIGHT) {
                Light* l = s->CreateObject<Light>();
                l->xVector = Get3(o.light_x);
                l->yVector = Get3(o.light_y);
                mesh = l;
            } else {
                mesh = s->CreateObject<TriangleMesh>();
            }
            mesh->position = Get3(o.position);
            CopyMaterialIn(o.material, mesh->material);
            mesh->triangles.reserve(o.triangle_count);
            for (uint32_t t = 0; t < o.triangle_count; ++t) {
                const mcrt_triangle& tr = d->triangles[o.first_triangle + t];
                vec3 a = Get3(tr.v0), b = Get3(tr.v1), c = Get3(tr.v2);
                mesh->triangles.push_back(Triangle{ a, b, c });
            }
        }
    }
    if (build_reference_octree) s->PrepareForRayTracing();
    else {
        std::vector<Object*>& lights = SceneAccess::Lights(*s);
        lights.clear();
        for (Object* o : SceneAccess::Objects(*s)) {
            Material& m = o->material;
            if (m.emission.r > 0.0 || m.emission.g > 0.0 || m.emission.b > 0.0) lights.push_back(o);
        }
        for (Object* o : SceneAccess::Objects(*s)) o->UpdateAABB();
    }
    RefScene* h = new RefScene;
    h->scene = s;
    return h;
}

void ref_scene_destroy(void* h)
{
    RefScene* r = static_cast<RefScene*>(h);
    if (!r) return;
    delete r->scene;
    delete r;
}

void ref_scene_counts(void* h, uint32_t* object_count, uint32_t* triangle_count, uint32_t* light_count)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    uint32_t tris = 0;
    for (Object* o : SceneAccess::Objects(s))
        if (TriangleMesh* m = dynamic_cast<TriangleMesh*>(o)) tris += uint32_t(m->triangles.size());
    *object_count = uint32_t(SceneAccess::Objects(s).size());
    *triangle_count = tris;
    *light_count = uint32_t(SceneAccess::Lights(s).size());
}

/* Flat dump; caller sized the arrays with ref_scene_counts. */
void ref_scene_dump(void* h, mcrt_object* objects, mcrt_triangle* triangles, double background[3])
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    background[0] = s.backgroundColor.r; background[1] = s.backgroundColor.g; background[2] = s.backgroundColor.b;
    uint32_t next = 0, idx = 0;
    for (Object* o : SceneAccess::Objects(s)) {
        mcrt_object& d = objects[idx++];
        std::memset(&d, 0, sizeof d);
        Put3(d.position, o->position);
        Put3(d.aabb_min, o->aabb.min);
        Put3(d.aabb_max, o->aabb.max);
        CopyMaterialOut(o->material, d.material);
        if (SphereObject* sp = dynamic_cast<SphereObject*>(o)) {
            d.kind = MCRT_OBJECT_SPHERE;
            d.radius = sp->radius;
        } else if (TriangleMesh* m = dynamic_cast<TriangleMesh*>(o)) {
            d.kind = MCRT_OBJECT_MESH;
            if (Light* l = dynamic_cast<Light*>(o)) {
                d.kind = MCRT_OBJECT_LIGHT;
                Put3(d.light_x, l->xVector);
                Put3(d.light_y, l->yVector);
            }
            d.first_triangle = next;
            d.triangle_count = uint32_t(m->triangles.size());
            for (Triangle& t : m->triangles) {
                mcrt_triangle& dt = triangles[next++];
                Put3(dt.v0, t.vertex0); Put3(dt.v1, t.vertex1); Put3(dt.v2, t.vertex2);
                Put3(dt.normal, t.normal);
            }
        }
    }
}

/* Indices (into the object list) of Scene::lights, in cache order. */
void ref_scene_lights(void* h, int32_t* out)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    auto& objs = SceneAccess::Objects(s);
    uint32_t k = 0;
    for (Object* l : SceneAccess::Lights(s)) {
        int32_t found = -1;
        for (size_t i = 0; i < objs.size(); ++i) if (objs[i] == l) found = int32_t(i);
        out[k++] = found;
    }
}

/* --------------------------------------------------------------- camera */
void* ref_camera_create(uint32_t w, uint32_t h, float fov_y, const float pos[3], const float target[3], const float up[3])
{
    CameraAccess* c = new CameraAccess(w, h, fov_y);
    c->SetView(Get3(pos), Get3(target), Get3(up));
    return c;
}
void ref_camera_destroy(void* c) { delete static_cast<CameraAccess*>(c); }

void ref_camera_dump(void* cv, mcrt_camera* out)
{
    CameraAccess& c = *static_cast<CameraAccess*>(cv);
    Put3(out->position, c.Position());
    out->fov_scale = c.FovScale();
    const glm::mat4& m = c.View();
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) out->view[col * 4 + row] = m[col][row];
    out->width = uint32_t(c.pixels.width());
    out->height = uint32_t(c.pixels.height());
}

void ref_camera_rays(void* cv, uint64_t n, const float* xs, const float* ys, float* out_o, float* out_d)
{
    CameraAccess& c = *static_cast<CameraAccess*>(cv);
    for (uint64_t i = 0; i < n; ++i) {
        Ray r = c.GetPixelRay(xs[i], ys[i]);
        Put3(out_o + 3 * i, r.origin);
        Put3(out_d + 3 * i, r.direction);
    }
}

/* ------------------------------------------------------------------- rng */
void ref_rng_doubles(uint64_t s0, uint64_t s1, uint64_t n, double* out)
{
    SeedableRng g; g.Seed(s0, s1);
    for (uint64_t i = 0; i < n; ++i) out[i] = g.RandomDouble();
}
void ref_rng_floats(uint64_t s0, uint64_t s1, uint64_t n, float* out)
{
    SeedableRng g; g.Seed(s0, s1);
    for (uint64_t i = 0; i < n; ++i) out[i] = g.RandomFloat();
}
void ref_rng_floats_range(uint64_t s0, uint64_t s1, float lo, float hi, uint64_t n, float* out)
{
    SeedableRng g; g.Seed(s0, s1);
    for (uint64_t i = 0; i < n; ++i) out[i] = g.RandomFloat(lo, hi);
}
void ref_sample_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t out[2]) { SampleSeed(seed, pixel, sample, out); }

/* ------------------------------------------------------------ intersection */
/* Scene::IntersectRay (source/scene.cpp:62-80) on n rays. */
void ref_intersect(void* h, uint64_t n, const float* o, const float* d, int32_t* out_obj, uint32_t* out_elem, float* out_t)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    auto& objs = SceneAccess::Objects(s);
    for (uint64_t i = 0; i < n; ++i) {
        Ray r(Get3(o + 3 * i), Get3(d + 3 * i));
        RayIntersectionInfo info;
        s.IntersectRay(r, info);
        int32_t idx = -1;
        if (info.object)
            for (size_t k = 0; k < objs.size(); ++k) if (objs[k] == info.object) { idx = int32_t(k); break; }
        out_obj[i] = idx;
        out_elem[i] = info.elementIndex;
        out_t[i] = info.hitDistance;
    }
}

/* ------------------------------------------------------------ unit pieces */
/* Material::BRDF (source/core/material.h:24-53) */
void ref_brdf(const mcrt_material* m, uint64_t n, const float* incident, const float* reflection, const float* normal, double* out)
{
    Material mat; CopyMaterialIn(*m, mat);
    for (uint64_t i = 0; i < n; ++i) {
        vec3 a = Get3(incident + 3 * i), b = Get3(reflection + 3 * i), c = Get3(normal + 3 * i);
        out[i] = mat.BRDF(a, b, c);
    }
}

/* Scene::RandomHemisphereRay (source/scene.cpp:11-30); fresh stream per call. */
void ref_hemisphere(uint64_t n, const float* normals, const uint64_t* states, float* out_dir, float* out_cos)
{
    Scene s;
    for (uint64_t i = 0; i < n; ++i) {
        SeedableRng g; g.Seed(states[2 * i], states[2 * i + 1]);
        vec3 o(0.0f), in(0.0f, 0.0f, -1.0f), nn = Get3(normals + 3 * i);
        float c = 0.0f;
        Ray r = SceneAccess::Hemisphere(s, o, in, nn, g, c);
        Put3(out_dir + 3 * i, r.direction);
        out_cos[i] = c;
    }
}

/* Object::GetSurfaceNormal */
void ref_surface_normal(void* h, uint64_t n, const int32_t* obj, const uint32_t* elem, const float* points, float* out)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    auto& objs = SceneAccess::Objects(s);
    for (uint64_t i = 0; i < n; ++i) Put3(out + 3 * i, objs[size_t(obj[i])]->GetSurfaceNormal(Get3(points + 3 * i), elem[i]));
}

/* Object::Intersects of ONE object through the virtual interface (source/objects/object.h:23): the bool it returns and
 * hitInfo as the call leaves it, starting from a fresh RayIntersectionInfo (source/core/ray.h:17-29) */
void ref_object_intersects(void* h, uint32_t index, uint64_t n, const float* o, const float* d, uint8_t* out_hit, uint32_t* out_elem, float* out_t)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    Object* ob = SceneAccess::Objects(s)[index];
    for (uint64_t i = 0; i < n; ++i) {
        RayIntersectionInfo info;
        out_hit[i] = ob->Intersects(Get3(o + 3 * i), Get3(d + 3 * i), info) ? 1 : 0;
        out_elem[i] = info.elementIndex;
        out_t[i] = info.hitDistance;
    }
}

/* Object::GetRandomPointOnSurface (source/objects/object.h:26-29, light.h:47-53, sphere.h:59-69): one point per generator
 * state, the advanced state written back */
void ref_object_random_points(void* h, uint32_t index, uint64_t n, uint64_t* states, float* out)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    Object* ob = SceneAccess::Objects(s)[index];
    SeedableRng g;
    for (uint64_t i = 0; i < n; ++i) {
        g.Seed(states[2 * i], states[2 * i + 1]);
        Put3(out + 3 * i, ob->GetRandomPointOnSurface(g));
        g.State(states + 2 * i);
    }
}

/* ----------------------------------------------------------------- tracing */
/* Scene::TraceRay (source/scene.cpp:94-234) on explicit rays, stream = states[2i..]. */
void ref_trace_rays(void* h, uint64_t n, const float* o, const float* d, const uint64_t* states, uint32_t depth, double* out)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    SeedableRng g;
    for (uint64_t i = 0; i < n; ++i) {
        g.Seed(states[2 * i], states[2 * i + 1]);
        ColorDbl c = s.TraceRay(Ray(Get3(o + 3 * i), Get3(d + 3 * i)), g, depth);
        out[3 * i] = c.r; out[3 * i + 1] = c.g; out[3 * i + 2] = c.b;
    }
}

/* One iteration of the spp loop of source/main.cpp:122-139 for listed
 * (pixel, sample) pairs; jitter and TraceRay share the per-sample stream. */
/* `g` is constructed once per thread: UniformRandomGenerator's ctor reads std::random_device
 * (source/core/randomization.cpp:8-13), far too slow to run per sample; its state is overwritten here. */
static inline ColorDbl OneSample(Scene& s, CameraAccess& c, SeedableRng& g, uint32_t pixel, uint32_t sample, uint64_t seed, uint32_t depth)
{
    uint64_t st[2];
    SampleSeed(seed, pixel, sample, st);
    g.Seed(st[0], st[1]);
    uint32_t w = uint32_t(c.pixels.width());
    uint32_t x = pixel % w, y = pixel / w;
    float sx = g.RandomFloat();
    float sy = g.RandomFloat();
    Ray ray = c.GetPixelRay(float(x) + sx, float(y) + sy);
    return s.TraceRay(ray, g, depth);
}

void ref_trace_samples(void* h, void* cv, uint64_t n, const uint32_t* pixels, const uint32_t* samples, uint64_t seed, uint32_t depth, double* out)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    CameraAccess& c = *static_cast<CameraAccess*>(cv);
    SeedableRng g;
    for (uint64_t i = 0; i < n; ++i) {
        ColorDbl v = OneSample(s, c, g, pixels[i], samples[i], seed, depth);
        out[3 * i] = v.r; out[3 * i + 1] = v.g; out[3 * i + 2] = v.b;
    }
}

/* Headless multithreaded render: threads chase the next free pixel through an
 * atomic counter as in source/main.cpp:89-95; sums of finite samples in
 * sample order go to out_sums (width*height*3 doubles, PixelBuffer layout
 * source/core/pixelbuffer.cpp:53-56).  Returns the number of dropped
 * (non-finite) samples. */
uint64_t ref_render(void* h, void* cv, uint32_t sample_begin, uint32_t sample_count, uint32_t depth, uint64_t seed,
                    uint32_t threads, double* out_sums)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    CameraAccess& c = *static_cast<CameraAccess*>(cv);
    const uint32_t npix = uint32_t(c.pixels.numPixels());
    if (threads == 0) threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    std::atomic<uint32_t> next{ 0 };
    std::atomic<uint64_t> dropped{ 0 };
    auto worker = [&]() {
        uint64_t myDropped = 0;
        SeedableRng rng;
        for (;;) {
            uint32_t p = next.fetch_add(1, std::memory_order_relaxed);
            if (p >= npix) break;
            double r = 0.0, g = 0.0, b = 0.0;
            for (uint32_t k = 0; k < sample_count; ++k) {
                ColorDbl v = OneSample(s, c, rng, p, sample_begin + k, seed, depth);
                if (Finite3(v)) { r += v.r; g += v.g; b += v.b; } else ++myDropped;
            }
            out_sums[3 * size_t(p)] += r; out_sums[3 * size_t(p) + 1] += g; out_sums[3 * size_t(p) + 2] += b;
        }
        dropped += myDropped;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return dropped.load();
}

/* Scene::TraceUnlit through pixel centres (source/main.cpp:126,135). */
void ref_render_unlit(void* h, void* cv, double* out)
{
    Scene& s = *static_cast<RefScene*>(h)->scene;
    CameraAccess& c = *static_cast<CameraAccess*>(cv);
    uint32_t w = uint32_t(c.pixels.width()), hh = uint32_t(c.pixels.height());
    for (uint32_t y = 0; y < hh; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            ColorDbl v = s.TraceUnlit(c.GetPixelRay(float(x) + 0.5f, float(y) + 0.5f));
            size_t i = 3 * (size_t(y) * w + x);
            out[i] = v.r; out[i + 1] = v.g; out[i + 2] = v.b;
        }
}

/* Tonemap + 8-bit store exactly as source/main.cpp:142-164 feeds
 * GLImageBuffer::SetPixel (source/opengl/data.cpp:28-50).  data.cpp itself
 * needs OpenGL headers, so its three arithmetic lines (clamp, *255.0,
 * GLubyte cast, BGRA order) are replayed here; the tonemap expressions use
 * glm exactly as main.cpp does. */
void ref_resolve(uint64_t npix, const double* sums, uint64_t spp, int tonemap, double gamma, double exposure, uint8_t* out_bgra, double* out_mean)
{
    for (uint64_t p = 0; p < npix; ++p) {
        ColorDbl outputColor = ColorDbl(sums[3 * p], sums[3 * p + 1], sums[3 * p + 2]) / double(spp);
        if (out_mean) { out_mean[3 * p] = outputColor.r; out_mean[3 * p + 1] = outputColor.g; out_mean[3 * p + 2] = outputColor.b; }
        if (tonemap == MCRT_TONEMAP_REINHARD) {
            outputColor = outputColor / (outputColor + ColorDbl(1.0));
            outputColor = pow(outputColor, ColorDbl(1.0 / gamma));
        } else if (tonemap == MCRT_TONEMAP_EXPOSURE) {
            outputColor = ColorDbl(1.0) - glm::exp(-outputColor * exposure);
            outputColor = pow(outputColor, ColorDbl(1.0 / gamma));
        }
        if (out_bgra) {
            double r = std::max(std::min(1.0, outputColor.r), 0.0);
            double g = std::max(std::min(1.0, outputColor.g), 0.0);
            double b = std::max(std::min(1.0, outputColor.b), 0.0);
            typedef unsigned char GLubyte;
            out_bgra[4 * p] = GLubyte(b * 255.0);
            out_bgra[4 * p + 1] = GLubyte(g * 255.0);
            out_bgra[4 * p + 2] = GLubyte(r * 255.0);
            out_bgra[4 * p + 3] = GLubyte(1.0 * 255.0);
        }
    }
}

} // extern "C"
