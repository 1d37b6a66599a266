// mcrt_scene.hpp - host-side scene description API, source-compatible with the
// reference renderer's public classes for the render path, backed by the HIP
// library through include/mcrt.h.
//
// What is mirrored (names, members, argument meaning, no-exception/return-false
// error behaviour) and where it comes from in the reference:
//   Material, SurfaceType, DiffuseType      source/core/material.h:8-19
//   Ray, RayIntersectionInfo                source/core/ray.h:8-29
//   AABB                                    source/core/aabb.h:11-28,98-113
//   Triangle                                source/core/triangle.h:8-23
//   UniformRandomGenerator                  source/core/randomization.h:13-50
//   Object, SphereObject, TriangleMesh,
//   Box, Light                              source/objects/*.h
//   PixelBuffer, Camera                     source/core/pixelbuffer.h, camera.h
//   Octree (Fill / maxCount / aabb)         source/accelerationstructures/octree.h:31-92
//   Scene, CornellBoxScene, HexagonScene    source/scene.h:16-92
//
// What is different on purpose: the classes DESCRIBE the scene; every piece of
// ray arithmetic (Intersects, BRDF, TraceRay recursion, hemisphere sampling)
// runs on the GPU.  Scene::IntersectRay / TraceRay / TraceUnlit keep their
// signatures and forward single queries to the device; the batch entry points
// (Scene::Render, Camera::Resolve) are what a driver should call.  There is no
// CPU fallback: without a device these calls report failure (false / black and
// a message on stderr), as the reference reports failure with false/nullptr.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mcrt.h"
#include "mcrt_math.hpp"

enum class SurfaceType { Diffuse, Specular, Diffuse_Specular, Refractive, COUNT };
enum class DiffuseType { Lambertian, OrenNayar, COUNT };

struct Material {
    ColorDbl color = ColorDbl{ 1.0, 1.0, 1.0 };
    ColorDbl emission = ColorDbl{ 0.0, 0.0, 0.0 };
    SurfaceType type = SurfaceType::Diffuse;
    DiffuseType diffuse = DiffuseType::Lambertian;
    float albedo = 1.0f;
    float roughness = 1.0f;
    float refractiveIndex = 1.52f;
    // source/core/material.h:24-53.  Evaluated on the device (mcrt_brdf, the kernels' own BRDF code) through the process-wide
    // context of MaterialDevice(); 0.0 and a message on stderr where there is no GPU.
    double BRDF(vec3& incident, vec3& reflection, vec3& normal);
};
// The GPU Material::BRDF runs on (a Material belongs to no scene): device index for the process-wide context, default 0.
void MaterialDevice(int deviceId);

struct Ray {
    vec3 origin;
    vec3 direction;
    Ray() = default;
    Ray(vec3 start, vec3 dir) : origin(start), direction(dir) {}
};

class Object;
struct RayIntersectionInfo {
    Object* object = nullptr;
    unsigned int elementIndex = 0;
    float hitDistance = FLOAT_INFINITY;
    void Reset() { object = nullptr; elementIndex = 0; hitDistance = FLOAT_INFINITY; }
};

struct AABB {
    vec3 center, min, max;
    AABB(vec3 centerPosition = vec3{ 0.0f }, vec3 dimensions = vec3{ 0.0f })
        : center(centerPosition), min(centerPosition - dimensions / 2.0f), max(centerPosition + dimensions / 2.0f) {}
    void Reset() { center = min = max = vec3{ 0.0f }; }
    void Encapsulate(const vec3& p);
    void Encapsulate(const AABB& other) { Encapsulate(other.min); Encapsulate(other.max); }
};

struct Triangle {
    vec3 vertex0, vertex1, vertex2, normal;
    Triangle(const vec3& v0, const vec3& v1, const vec3& v2);
};

// xorshift128+.  The draws of a PATH happen on the device: the host object carries the 128-bit state in and out of
// Scene::TraceRay.  The host-side draws below serve the driver's own generator - the pixel picks of the random-pixel
// scheduler, source/main.cpp:27,78-79 - with the reference's arithmetic (source/core/randomization.h:23-43, .cpp:15-33).
class UniformRandomGenerator {
public:
    uint64_t xorseed[2] = { 0, 0 };
    UniformRandomGenerator();                                     // std::random_device, like the reference
    UniformRandomGenerator(uint64_t a, uint64_t b) { xorseed[0] = a; xorseed[1] = b; }
    // state for camera sample (seed, pixel, sample): the counter mapping of mcrt_rng_seed
    static UniformRandomGenerator ForSample(uint64_t seed, uint32_t pixel, uint32_t sample);
    double RandomDouble();
    double RandomDouble(double min, double max) { return min + (max - min) * RandomDouble(); }
    float RandomFloat() { return float(RandomDouble()); }
    float RandomFloat(float min, float max) { return min + (max - min) * float(RandomDouble()); }
};

class Scene;
class Object {
protected:
    friend class Scene;
    Scene* scene = nullptr;            // the scene that created the object (Scene::CreateObject) and its place in the object order
    unsigned int sceneIndex = 0;
public:
    vec3 position;
    Material material;
    float area = 1.0f;
    AABB aabb;
    Object() = default;
    virtual ~Object() = default;
    // The three ray-arithmetic virtuals of source/objects/object.h:23-29.  They run on the device against the scene image
    // uploaded by the last Scene::PrepareForRayTracing (mcrt_object_intersect / mcrt_surface_normals /
    // mcrt_random_surface_points): an object edited since then answers with its old geometry, as the reference's objects
    // answer with a stale AABB until PrepareForRayTracing has run.  An object that belongs to no prepared scene reports
    // a miss / a zero vector and a message on stderr.
    virtual bool Intersects(vec3 rayOrigin, vec3 rayDirection, RayIntersectionInfo& hitInfo);
    virtual vec3 GetSurfaceNormal(vec3 location, unsigned int index);
    virtual bool IsLight() { return false; }
    virtual vec3 GetRandomPointOnSurface(UniformRandomGenerator& gen);
    virtual double PDF() { return 1.0 / area; }
    virtual void UpdateAABB() {}
    virtual int Kind() const = 0;                                  // MCRT_OBJECT_*
};

class SphereObject : public Object {
public:
    float radius = 1.0f;
    void UpdateAABB() override { aabb = AABB(position, vec3{ radius }); }   // half-size on purpose, sphere.h:71-74
    int Kind() const override { return MCRT_OBJECT_SPHERE; }
};

class TriangleMesh : public Object {
public:
    std::vector<Triangle> triangles;
    void AddQuad(vec3 p1, vec3 p2, vec3 p3, vec3 p4);              // ccw with respect to the normal
    void UpdateAABB() override;
    // Index-correct OBJ reader (v / f records; faces triangulated the way the reference's OBJ library does it), offset by `position`.
    bool LoadMesh(std::string path);
    int Kind() const override { return MCRT_OBJECT_MESH; }
};

class Box : public TriangleMesh {
public:
    void SetGeometry(vec3 basePosition, vec3 upVector, vec3 sideVector, float width, float depth, float height);
};

class Light : public TriangleMesh {
public:
    vec3 normal, xVector, yVector;
    Light(ColorDbl lightEmission = ColorDbl{ 1.0 }) { material.emission = lightEmission; }
    void SetGeometry(vec3 centerPosition, vec3 lightDirection, vec3 sideDirection, vec2 quadDimensions);
    bool IsLight() override { return true; }
    int Kind() const override { return MCRT_OBJECT_LIGHT; }
};

class PixelBuffer {
protected:
    unsigned int dataSize = 0, imageWidth = 0, imageHeight = 0;
    double dx = 0.0, dy = 0.0, aspect = 1.0;
    std::vector<double> data;                                      // accumulator sums: Scene::ReadBack and Accumulate
    uint64_t rays = 0;                                             // samples every pixel has from Scene::Render passes
    std::vector<uint64_t> rayCount;                                // per pixel, on top of `rays`: samples added one by one (Accumulate)
    friend class Camera;
    friend class Scene;
public:
    PixelBuffer(unsigned int width, unsigned int height);
    int size() const { return int(dataSize); }
    int numPixels() const { return int(imageWidth * imageHeight); }
    int width() const { return int(imageWidth); }
    int height() const { return int(imageHeight); }
    double deltaX() const { return dx; }
    double deltaY() const { return dy; }
    double aspectRatio() const { return aspect; }
    unsigned int PixelArrayIndex(unsigned int x, unsigned int y) { return y * imageWidth * 3 + x * 3; }
    // source/core/pixelbuffer.cpp:39-62 (the reference indexes its ray counts with the colour index, :45,:58-61)
    void Accumulate(unsigned int x, unsigned int y, ColorDbl color);
    uint64_t GetRayCount(unsigned int x, unsigned int y) { return rays + (rayCount.empty() ? 0u : rayCount[size_t(y) * imageWidth + x]); }
    ColorDbl GetPixelColor(unsigned int x, unsigned int y);
};

class Camera {
protected:
    mcrt_camera view{};
    float fov = 90.0f;
    friend class Scene;
public:
    PixelBuffer pixels;
    Camera(unsigned int width, unsigned int height, float fovY);
    void SetView(vec3 position, vec3 lookAtPosition, vec3 cameraUp = vec3{ 0.0f, 1.0f, 0.0f });
    // source/core/camera.h:36-57, on the host, operation by operation (the kernels' camera_ray is the same sequence)
    Ray GetPixelRay(float x, float y) const;
    const mcrt_camera& Description() const { return view; }
};

// Scene::octree: the entry point name and knobs of the reference's Octree; the structure itself is the
// flat per-primitive octree built inside mcrt_upload_scene.
class Octree {
public:
    unsigned int maxCount = 0;         // 0 = library default
    unsigned int maxDepth = 0;
    // true: PrepareForRayTracing has the GPU build the tree (mcrt_upload_scene_device_build) - for scenes edited between frames,
    // what the reference's incremental InsertIfOverlaps is for (octree.h:50-70): edit the objects, call PrepareForRayTracing again
    bool buildOnDevice = false;
    float lastBuildMs = 0.0f;          // device time of the last device build
    AABB aabb;                         // root box after Fill
    unsigned int nodeCount = 0, primitiveRefs = 0, depth = 0;
    bool ldsResident = false;
    void PrintDebug() const;
};

class Scene {
protected:
    std::vector<Object*> objects;
    std::vector<Object*> lights;
    mcrt_ctx* ctx = nullptr;
    int device = 0;
    bool prepared = false;
    std::vector<mcrt_object> flatObjects;
    std::vector<mcrt_triangle> flatTriangles;
    const Camera* boundCamera = nullptr;
    bool EnsureContext();
    bool Fail(const char* what) const;
    bool Bind(Camera& camera);                                       // mcrt_set_camera once per (scene, camera)
public:
    Octree octree;
    ColorDbl backgroundColor = { 0.0, 0.0, 0.0 };

    Scene() = default;
    virtual ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    template <class T> T* CreateObject()
    {
        T* newObject = new T();
        newObject->scene = this;
        newObject->sceneIndex = (unsigned int)objects.size();
        objects.push_back(newObject);
        prepared = false;
        return newObject;
    }

    // objects in creation order - the order of the nearest-hit tie-break (the reference keeps `objects` protected, source/scene.h:18-20)
    unsigned int ObjectCount() const { return (unsigned int)objects.size(); }
    Object* GetObject(unsigned int index) { return objects[index]; }

    void UseDevice(int deviceId) { device = deviceId; }
    mcrt_ctx* Context() { return EnsureContext() ? ctx : nullptr; }

    // Light cache + UpdateAABB (source/scene.cpp:41-55), then flatten and upload (replaces octree.Fill, :58).
    void PrepareForRayTracing();
    // After PrepareForRayTracing: the listed objects (indices into `objects`) MOVED - same kinds, same triangle counts.  The tree is
    // kept and refitted on the device when every moved primitive is still inside the cells that reference it, rebuilt on the device
    // otherwise (mcrt_refit_scene; the reference's incremental entry point is Octree::InsertIfOverlaps, octree.h:50-70).
    // *rebuilt (may be null) tells which; octree.lastBuildMs holds the device time of whichever ran.
    bool ObjectsMoved(const std::vector<unsigned int>& indices, bool* rebuilt = nullptr);
    // The flat description PrepareForRayTracing uploads (no device needed).
    bool Flatten(mcrt_scene_desc& out);

    bool IntersectRay(Ray& ray, RayIntersectionInfo& hitInfo) const;      // source/scene.h:41
    ColorDbl TraceUnlit(Ray ray) const;                                    // source/scene.h:43
    double MaxImportance(ColorDbl& importance)
    {
        double yz = importance.g < importance.b ? importance.b : importance.g;
        return importance.r < yz ? yz : importance.r;
    }
    ColorDbl TraceRay(Ray ray, UniformRandomGenerator& uniformGenerator, unsigned int traceDepth = 5, ColorDbl importance = ColorDbl{ 1.0 });

    // Batch entry points (what replaces TracePixels, source/main.cpp:169-184).
    bool Render(Camera& camera, uint32_t sampleBegin, uint32_t sampleCount, unsigned int traceDepth, uint64_t seed);
    bool ReadBack(Camera& camera);                                   // device accumulators -> camera.pixels
    // camera.pixels (sums of `samplesDone` samples per pixel, e.g. from a checkpoint) -> device accumulators
    bool Restore(Camera& camera, const std::vector<double>& sums, uint64_t samplesDone);
    bool Resolve(Camera& camera, int tonemap, double gamma, double exposure, std::vector<uint8_t>& bgra8);
    bool Clear(Camera& camera);

    // Per-sample radiance of an explicit (pixel, sample index) list: what one RayTraceNextPixel call computes for a
    // randomly picked pixel (source/main.cpp:70-86,122-139); the caller adds the values with PixelBuffer::Accumulate.
    bool TraceSamples(Camera& camera, const std::vector<uint32_t>& pixels, const std::vector<uint32_t>& samples, uint64_t seed,
                      unsigned int traceDepth, std::vector<double>& radiance);
    // camera.pixels (sums and per-pixel ray counts filled by PixelBuffer::Accumulate) -> 8-bit image: the mean of every
    // pixel over ITS OWN ray count (source/main.cpp:142), pixels never picked stay black
    bool ResolvePerPixelCounts(Camera& camera, int tonemap, double gamma, double exposure, std::vector<uint8_t>& bgra8);

    // Several GPUs (or several contexts of one GPU), one Scene + Camera each, every one rendering its own share of the
    // sample indices - the worker threads of source/main.cpp:59-62,250-262 turned into one context per device.
    // Reduce: sum of all ranks' accumulators into `root` (mcrt_reduce, RCCL; every rank calls it from its own thread).
    // AddAccumulatorOf: this += other's accumulator, for contexts that share this GPU (no communicator needed).
    // Both leave camera.pixels' sample count to the caller: SetSampleCount tells Resolve what the sums now hold.
    bool Reduce(Camera& camera, void* ncclComm, int root);
    bool AddAccumulatorOf(Camera& camera, Scene& other);
    void SetSampleCount(Camera& camera, uint64_t samples) { camera.pixels.rays = samples; }

    virtual void MoveCameraToRecommendedPosition(Camera& camera);

    // what Object::Intersects / GetSurfaceNormal / GetRandomPointOnSurface forward to (object `index` of the uploaded image)
    bool ObjectIntersects(unsigned int index, vec3 rayOrigin, vec3 rayDirection, unsigned int& elementIndex, float& hitDistance) const;
    bool ObjectSurfaceNormal(unsigned int index, vec3 location, unsigned int element, vec3& normal) const;
    bool ObjectRandomPoint(unsigned int index, UniformRandomGenerator& gen, vec3& point) const;
};

// source/opengl/screenshot.cpp:13-76 without the window: the image is the resolved BGRA8 buffer (Scene::Resolve, top row first)
// instead of the GL framebuffer.  TakeScreenshot keeps the reference's file naming - extension stripped, a five-digit counter
// appended, the first name that does not exist yet ("shot00001.png", "shot00002.png", ...) - and returns the name written
// (empty on failure).  WritePNG: 8-bit RGBA, one IDAT of stored (uncompressed) deflate blocks: any PNG reader opens it.
bool WritePNG(const std::string& path, const std::vector<uint8_t>& bgra8, unsigned width, unsigned height);
std::string TakeScreenshot(std::string filename, const std::vector<uint8_t>& bgra8, unsigned screenWidth, unsigned screenHeight);

class HexagonScene : public Scene {
protected:
    TriangleMesh *ceiling, *floor, *walls1, *walls2, *walls3;
public:
    HexagonScene();
    void MoveCameraToRecommendedPosition(Camera& camera) override;
    void AddExampleObjects(float radius = 1.5f);
    void AddExampleLight(ColorDbl lightColor, bool usePoint = false);
};

class CornellBoxScene : public Scene {
protected:
    float halfLength = 0.0f, halfWidth = 0.0f, halfHeight = 0.0f;
    TriangleMesh *leftWall, *rightWall, *whiteSegments;
public:
    CornellBoxScene(float length, float width, float height);
    void MoveCameraToRecommendedPosition(Camera& camera) override;
    void AddExampleObjects(float radius = 1.5f);
    void AddExampleLight(ColorDbl lightColor, bool usePoint = false);
};
