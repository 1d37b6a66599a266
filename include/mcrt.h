/*
 * mcrt.h - C ABI of the MI355X path-tracing backend (libmcrt_hip.so).
 *
 * This is the drop-in boundary for the reference renderer's CPU per-pixel
 * render loop.  The reference (DennyLindberg/Monte-Carlo-Ray-Tracer) has no
 * FFI of its own; the calls below replace, one for one, what its worker
 * threads do through the public C++ class API.  Every entry point cites the
 * reference interface it stands in for as `source/<file>:<line>` (paths
 * relative to the reference checkout).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++ or torch types cross it;
 *   - every function returns an int status (MCRT_OK == 0) and never throws;
 *   - the library never takes ownership of caller memory;
 *   - one mcrt_ctx is used from one host thread at a time; multi-GPU means
 *     one context (and normally one process) per device;
 *   - there is NO CPU fallback: without a HIP device mcrt_create fails.
 */
#ifndef MCRT_H
#define MCRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status */
#define MCRT_OK 0
#define MCRT_ERR_INVALID 1     /* bad argument / bad scene description      */
#define MCRT_ERR_NO_DEVICE 2   /* no HIP device, or device_id out of range  */
#define MCRT_ERR_HIP 3         /* a HIP runtime call failed                 */
#define MCRT_ERR_STATE 4       /* call order violated (no scene/camera yet) */
#define MCRT_ERR_LIMIT 5       /* a documented size limit was exceeded      */

/* ------------------------------------------------------- scene description
 * POD mirror of the reference's scene graph after
 * Scene::PrepareForRayTracing (source/scene.cpp:38-59).  Objects keep their
 * insertion order because that order is the nearest-hit tie-break order
 * (source/scene.cpp:71-76). */

/* SurfaceType / DiffuseType, source/core/material.h:8-9 */
enum { MCRT_SURFACE_DIFFUSE = 0, MCRT_SURFACE_SPECULAR = 1,
       MCRT_SURFACE_DIFFUSE_SPECULAR = 2, MCRT_SURFACE_REFRACTIVE = 3 };
enum { MCRT_DIFFUSE_LAMBERTIAN = 0, MCRT_DIFFUSE_OREN_NAYAR = 1 };

/* struct Material, source/core/material.h:11-19 */
typedef struct mcrt_material {
    double color[3];
    double emission[3];
    int32_t surface;           /* MCRT_SURFACE_*  */
    int32_t diffuse;           /* MCRT_DIFFUSE_*  */
    float albedo;
    float roughness;
    float refractive_index;
    int32_t reserved;
} mcrt_material;

/* struct Triangle, source/core/triangle.h:8-23 (three vertices + the face
 * normal its constructor derives). */
typedef struct mcrt_triangle {
    float v0[3];
    float v1[3];
    float v2[3];
    float normal[3];
} mcrt_triangle;

/* Which concrete reference class an object is. */
enum {
    MCRT_OBJECT_SPHERE = 0,    /* SphereObject, source/objects/sphere.h:8   */
    MCRT_OBJECT_MESH = 1,      /* TriangleMesh / Box, source/objects/mesh.h */
    MCRT_OBJECT_LIGHT = 2      /* Light (quad emitter), objects/light.h:8   */
};

/* class Object and its subclasses, source/objects/object.h:12-33. */
typedef struct mcrt_object {
    int32_t kind;              /* MCRT_OBJECT_*                             */
    uint32_t first_triangle;   /* into mcrt_scene_desc.triangles (meshes)   */
    uint32_t triangle_count;
    float radius;              /* SphereObject::radius                      */
    float position[3];         /* Object::position                          */
    float aabb_min[3];         /* Object::aabb after UpdateAABB (meshes):   */
    float aabb_max[3];         /*   the per-mesh gate of objects/mesh.cpp:11 */
    float light_x[3];          /* Light::xVector (half extent)              */
    float light_y[3];          /* Light::yVector (half extent)              */
    mcrt_material material;
} mcrt_object;

typedef struct mcrt_scene_desc {
    uint32_t object_count;
    uint32_t triangle_count;
    const mcrt_object* objects;
    const mcrt_triangle* triangles;
    double background[3];      /* Scene::backgroundColor, source/scene.h:26 */
} mcrt_scene_desc;

/* class Camera, source/core/camera.h:9-58, after SetView. */
typedef struct mcrt_camera {
    float position[3];
    float fov_scale;           /* Camera::fovPixelScale                     */
    float view[16];            /* inverse(lookAtRH), column-major like glm  */
    uint32_t width;
    uint32_t height;
} mcrt_camera;

/* How nearest-hit queries walk the scene.  Results are identical by
 * construction.
 *   OCTREE  per-lane stack traversal of the flat octree (any scene size);
 *   BRUTE   source/scene.cpp:62-80 restated literally (on-device cross-check);
 *   SCAN    wave-uniform walk of the object list with per-mesh gates, geometry
 *           through scalar loads - the fast path for scenes of a few hundred
 *           primitives, where 64 lanes on 64 different tree paths lose to one
 *           lock-step pass;
 *   FLAT    SCAN with the work dealt out: a mesh most of the wavefront needs
 *           (>= 24 lanes pass its gate) is walked in lock step as in SCAN; for
 *           the others the (ray, triangle) pairs that need testing are laid
 *           end to end and dealt evenly to the 64 lanes through LDS, results
 *           merged per ray with a 64-bit atomic min on (t, object, element) -
 *           the reference's tie-break order.  LDS-resident scenes only;
 *   AUTO    FLAT when the scene has at most MCRT_SCAN_MAX_PRIMS primitives,
 *           OCTREE otherwise (default). */
enum { MCRT_ACCEL_OCTREE = 0, MCRT_ACCEL_BRUTE = 1, MCRT_ACCEL_SCAN = 2, MCRT_ACCEL_AUTO = 3, MCRT_ACCEL_FLAT = 4 };
#define MCRT_SCAN_MAX_PRIMS 256
#define MCRT_SCAN_PAIR_LIMIT 65536   /* scenes with more triangles carry no scan records: SCAN is refused */

typedef struct mcrt_accel_params {
    uint32_t max_leaf_prims;   /* Octree::maxCount analogue (octree.h:31); 0 = default (4; 8 for scenes beyond 1024 primitives) */
    uint32_t max_depth;        /* 0 = default                               */
    float pad_rel;             /* node box padding, fraction of root extent; 0 = default */
} mcrt_accel_params;

/* Tonemap switch of source/main.cpp:45-48,144-164. */
enum { MCRT_TONEMAP_NONE = 0, MCRT_TONEMAP_REINHARD = 1, MCRT_TONEMAP_EXPOSURE = 2 };

typedef struct mcrt_ctx mcrt_ctx;

/* ------------------------------------------------------------- life cycle */

/* Bind a context to HIP device `device_id`.  Replaces the implicit "this
 * process renders on the CPU" of source/main.cpp:220-262. */
int mcrt_create(int device_id, mcrt_ctx** out_ctx);
int mcrt_destroy(mcrt_ctx* ctx);

/* Text of the last failure on this context (or of the last failed
 * mcrt_create when ctx is NULL).  The reference reports failure as
 * false/nullptr only (source/scene.cpp:79). */
const char* mcrt_last_error(const mcrt_ctx* ctx);

/* Launch all later work of this context on the caller's hipStream_t
 * (NULL = the context's own stream).  No reference counterpart. */
int mcrt_set_stream(mcrt_ctx* ctx, void* hip_stream);

/* ------------------------------------------------------------ scene/camera */

/* Flatten + upload the scene: Scene::PrepareForRayTracing,
 * source/scene.cpp:38-59 (light cache :41-49; the octree of
 * source/accelerationstructures/octree.h:77-92 is rebuilt here as a flat,
 * per-primitive structure of node and leaf records).  `accel` may be NULL.
 * Limits (MCRT_ERR_LIMIT): at most 512 objects, and their table has to fit the
 * LDS budget; at most 8388607 primitives (spheres + triangles: the length of a
 * node's list is a 23-bit field of its record); fewer than 2^24 octree nodes.
 * accel->max_depth above
 * MCRT_MAX_TREE_DEPTH (12) is clamped to it, not refused: deeper levels only
 * split cells that are already smaller than 1/4096 of the scene. */
#define MCRT_MAX_TREE_DEPTH 12
int mcrt_upload_scene(mcrt_ctx* ctx, const mcrt_scene_desc* scene,
                      const mcrt_accel_params* accel);

/* The same upload with the octree built ON THE DEVICE: for scenes edited between
 * frames - objects moved, triangles added or removed - where rebuilding on the host
 * (1.7 s for 10^6 triangles) would stall the render loop.  This is the build's
 * counterpart of the reference's incremental Octree::InsertIfOverlaps / Fill
 * (source/accelerationstructures/octree.h:50-70,77-92): after an edit the caller
 * passes the whole edited scene again and kernels re-index it level by level (same
 * cells, split rule, overlap tests and node numbering as the host build, so every
 * query answers as before: the brute force of source/scene.cpp:62-80).
 * mcrt_last_build_ms: device time of that build, for the scene in use
 * (MCRT_ERR_STATE when the scene in use was built on the host, or none is). */
int mcrt_upload_scene_device_build(mcrt_ctx* ctx, const mcrt_scene_desc* scene,
                                   const mcrt_accel_params* accel);
int mcrt_last_build_ms(mcrt_ctx* ctx, float* out_ms);

/* The cheap half of an edit: primitives that only MOVED since the upload - the same
 * objects with the same kinds and triangle counts; `scene` is the whole edited scene,
 * `moved_objects` the indices of the objects whose geometry changed (materials may
 * change anywhere).  When every primitive of those objects is still referenced
 * wherever the builder's rule wants a reference (a move inside the cells that already
 * hold it), the tree stays: kernels rewrite the triangle data inlined in the leaf
 * records and refit every node's tight box bottom-up (a few passes over the node
 * array, well under a millisecond for 10^6 triangles).  Otherwise - a primitive left
 * its cells, or the scene is not the same set of objects - the call falls back to
 * mcrt_upload_scene_device_build with the parameters of the tree in use and sets
 * *out_rebuilt = 1 (may be NULL).  Either way queries answer as the brute force of
 * source/scene.cpp:62-80 on the edited scene.  The reference's counterpart is its
 * incremental Octree::InsertIfOverlaps (source/accelerationstructures/octree.h:50-70).
 * A primitive listed as not moved that did move is the caller's error (its old
 * references are trusted).  After a refit that kept the tree mcrt_last_build_ms reports
 * the refit's kernel time. */
int mcrt_refit_scene(mcrt_ctx* ctx, const mcrt_scene_desc* scene, uint32_t moved_count,
                     const uint32_t* moved_objects, int* out_rebuilt);

/* Camera::Camera + Camera::SetView, source/core/camera.h:19-34; also sizes
 * and zeroes the PixelBuffer accumulators (source/core/pixelbuffer.cpp:7-23):
 * the accumulator in use is cleared, a caller-bound one included.  When the
 * image SIZE changes, a buffer bound with mcrt_bind_accumulator no longer fits:
 * the context returns to its own accumulator and the caller binds a buffer of
 * the new size again (mcrt_accumulator tells which one is in use). */
int mcrt_set_camera(mcrt_ctx* ctx, const mcrt_camera* camera);

/* Host-only helper: fill an mcrt_camera the way the reference does -
 * Camera::Camera (fovPixelScale = tan(fovY/2), source/core/camera.h:19-26) and
 * Camera::SetView (inverse(glm::lookAtRH), :30-34). */
int mcrt_camera_look_at(uint32_t width, uint32_t height, float fov_y_degrees, const float position[3],
                        const float look_at[3], const float up[3], mcrt_camera* out);

/* Choose MCRT_ACCEL_AUTO (default), _OCTREE, _SCAN or _BRUTE for later calls. */
int mcrt_set_accel(mcrt_ctx* ctx, int accel_mode);

/* ------------------------------------------------------------ parity paths */

/* Camera::GetPixelRay(x, y), source/core/camera.h:36-57, for n pixel
 * positions.  Host buffers; out arrays are n*3 floats. */
int mcrt_camera_rays(mcrt_ctx* ctx, uint64_t n, const float* xs, const float* ys,
                     float* out_origins, float* out_dirs);

/* Scene::IntersectRay, source/scene.cpp:62-80, for n rays (host buffers,
 * n*3 floats each).  out_obj = object index or -1 on a miss, out_elem =
 * RayIntersectionInfo::elementIndex, out_t = hitDistance (FLT_MAX on a miss),
 * source/core/ray.h:17-29.  Directions are unit vectors, as every ray the
 * reference builds is (camera.h:56, scene.cpp:11-30); zero, NaN and infinite
 * components answer as the reference's arithmetic does.  A direction of another
 * length is outside the contract: SphereObject::Intersects (sphere.h:16-52)
 * then reports distances that are not along the ray, which only MCRT_ACCEL_BRUTE
 * reproduces.
 * Origins may lie anywhere.  MCRT_ACCEL_OCTREE answers a ray whose origin has a
 * coordinate beyond mcrt_octree_info.far_origin_limit (= pad / (32 * FLT_EPSILON),
 * about 26 scene sizes from the coordinate origin with the default padding) - or
 * not a number - with the reference's loop over all objects instead of the tree:
 * out there one ulp of the origin's coordinate is no longer small against the
 * padding of the tree, and the reference's own arithmetic decides inside that
 * noise.  Same results, O(primitives) per such ray. */
int mcrt_intersect(mcrt_ctx* ctx, uint64_t n, const float* origins, const float* dirs,
                   int32_t* out_obj, uint32_t* out_elem, float* out_t);

/* Same, all five arrays already in device memory (no copies, no sync). */
int mcrt_intersect_device(mcrt_ctx* ctx, uint64_t n, const float* d_origins,
                          const float* d_dirs, int32_t* d_obj, uint32_t* d_elem,
                          float* d_t);

/* mcrt_intersect for callers that must not allocate device memory (code that
 * runs under HIP graph capture next to it, tight serving loops, harnesses that
 * forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments, same
 * results as mcrt_intersect, but the device staging of the five host arrays
 * lives in a static device array of the library (MCRT_NOALLOC_MAX_RAYS * 36
 * bytes, 14.4 MB, held by every GPU the library is used on until the process
 * ends) instead of hipMalloc'ed buffers: the call makes no hipMalloc / hipFree.
 * n > MCRT_NOALLOC_MAX_RAYS is refused with MCRT_ERR_LIMIT.  All contexts on
 * one GPU share that GPU's staging array: calls that overlap in time on one
 * GPU (two contexts, two threads) race - serialise them.  The array keeps the
 * previous call's bytes; nothing in it needs clearing (every output of rays
 * [0, n) is written before it is copied back). */
#define MCRT_NOALLOC_MAX_RAYS 400000u
int mcrt_intersect_noalloc(mcrt_ctx* ctx, uint64_t n, const float* origins, const float* dirs,
                           int32_t* out_obj, uint32_t* out_elem, float* out_t);

/* ---- the per-object virtuals of class Object, source/objects/object.h:23-29,
 * and Material::BRDF, on batches (host buffers).  The host mirror's
 * Object::Intersects / GetSurfaceNormal / GetRandomPointOnSurface and
 * Material::BRDF forward single calls to these.
 *
 * mcrt_object_intersect: Object::Intersects(rayOrigin, rayDirection, hitInfo)
 * (object.h:23) of object `object_index` of the uploaded scene alone -
 * SphereObject::Intersects (sphere.h:16-52) or TriangleMesh::Intersects
 * (mesh.cpp:9-41: AABB gate, then every triangle, the first of the nearest
 * wins) - for n rays.  out_hit[i] = 1 / 0 (the bool it returns), out_elem /
 * out_t = hitInfo.elementIndex / hitDistance after the call (0 / FLT_MAX on a
 * miss: RayIntersectionInfo::Reset, ray.h:24-29).
 * mcrt_surface_normals: Object::GetSurfaceNormal(location, index) (object.h:24;
 * sphere.h:54-57 normalize(location - position), mesh.cpp:43-46 the stored face
 * normal) for n (object, element, location) triples; out = n*3 floats.
 * mcrt_random_surface_points: Object::GetRandomPointOnSurface(gen)
 * (object.h:26-29 the position; light.h:47-53 the quad's central quarter;
 * sphere.h:59-69) - one point of object `object_index` per generator state;
 * rng_states: n*2 uint64 in, advanced on return (0 or 2 draws each).
 * mcrt_brdf: Material::BRDF(incident, reflection, normal),
 * source/core/material.h:24-53 (Lambertian: albedo; Oren-Nayar with the
 * kernels' fixed-order trigonometry, within 1 ulp of libm's), n direction
 * triples; out = n doubles.  Needs no scene. */
int mcrt_object_intersect(mcrt_ctx* ctx, uint32_t object_index, uint64_t n, const float* origins,
                          const float* dirs, uint8_t* out_hit, uint32_t* out_elem, float* out_t);
int mcrt_surface_normals(mcrt_ctx* ctx, uint64_t n, const uint32_t* objects, const uint32_t* elems,
                         const float* positions, float* out_normals);
int mcrt_random_surface_points(mcrt_ctx* ctx, uint32_t object_index, uint64_t n, uint64_t* rng_states,
                               float* out_points);
int mcrt_brdf(mcrt_ctx* ctx, const mcrt_material* material, uint64_t n, const float* incident,
              const float* reflection, const float* normal, double* out);

/* Per-sample radiance, i.e. one iteration of the spp loop of
 * source/main.cpp:122-139 without the Accumulate: jitter, GetPixelRay,
 * Scene::TraceRay (source/scene.cpp:94-234) with max_depth.  Sample k uses
 * pixel index pixels[k] (= y*width + x) and sample index samples[k]; its
 * random stream is mcrt_rng_seed(seed, pixel, sample).  out = n*3 doubles. */
int mcrt_trace_samples(mcrt_ctx* ctx, uint64_t n, const uint32_t* pixels,
                       const uint32_t* samples, uint64_t seed, uint32_t max_depth,
                       double* out_radiance);

/* Scene::TraceRay(ray, uniformGenerator, traceDepth, importance),
 * source/scene.cpp:94-234 (signature source/scene.h:50), for n explicit rays.
 * rng_states holds n*2 uint64: the xorshift128+ state of each ray's generator
 * on entry and, like the reference's by-reference generator, the advanced
 * state on return.  importance: n*3 doubles or NULL for (1,1,1).
 * out = n*3 doubles.  Host buffers. */
int mcrt_trace_rays(mcrt_ctx* ctx, uint64_t n, const float* origins, const float* dirs,
                    uint64_t* rng_states, uint32_t max_depth, const double* importance,
                    double* out_radiance);

/* Scene::TraceUnlit, source/scene.cpp:82-92, through pixel centres
 * (source/main.cpp:126).  out = width*height*3 doubles. */
int mcrt_render_unlit(mcrt_ctx* ctx, double* out_color);

/* ----------------------------------------------------------------- render */

/* Initial state of the xorshift128+ generator (source/core/randomization.h:
 * 35-43) for one camera sample: the counter (seed, pixel, sample) hashed to
 * 128 bits.  Pure host function; the kernels use the same mapping. */
void mcrt_rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t out_state[2]);

/* Zero the radiance accumulators: PixelBuffer ctor loop,
 * source/core/pixelbuffer.cpp:14-18. */
int mcrt_clear(mcrt_ctx* ctx);

/* Samples dropped because a component was not finite (stated deviation, see
 * mcrt_render) since the last mcrt_clear / mcrt_set_camera, over all passes.
 * The reference would have added them and poisoned the pixel for good
 * (source/core/pixelbuffer.cpp:39-46); mcrt_resolve still divides by the
 * spp_total it is given, as the reference divides by its ray count.
 * Synchronises the context's stream. */
int mcrt_dropped_samples(mcrt_ctx* ctx, uint64_t* out_count);

/* Use caller-owned device memory (width*height*3 doubles) as the radiance
 * accumulator, so the caller can all-reduce it (RCCL) between passes.
 * NULL returns to the context's own buffer.  No reference counterpart. */
int mcrt_bind_accumulator(mcrt_ctx* ctx, double* d_accum);

/* Device address of the accumulator currently in use. */
int mcrt_accumulator(mcrt_ctx* ctx, double** out_d_accum);

/* ---- multi-GPU: the reduce() of the drop-in boundary.
 * The reference runs one worker thread per hardware thread over shared
 * accumulators (source/main.cpp:59-62,250-262).  Here every GPU has its own
 * context and accumulator and renders its own share of the sample indices;
 * between passes the accumulators are summed into one rank's, which resolves.
 *
 * mcrt_reduce: ncclReduce(sum, ncclDouble, width*height*3) of the accumulator
 * in use into rank `root`, in place, on the context's stream (asynchronous).
 * `comm` is an ncclComm_t (RCCL); every rank of the communicator calls it, each
 * from its own thread or process.  RCCL is bound at run time (symbols already
 * in the process, else librccl.so.1 / librccl.so, or the library named by the
 * environment variable MCRT_RCCL_LIB): without a loadable librccl that reports
 * an NCCL 2.x API these three calls fail with MCRT_ERR_NO_DEVICE and a message,
 * and nothing else is affected.
 * mcrt_comm_init_all / mcrt_comm_destroy: ncclCommInitAll / ncclCommDestroy for
 * the one-process-many-GPUs case (out_comms: n ncclComm_t, rank i on
 * devices[i]), so that a C caller needs no RCCL headers.  A process that
 * already has communicators (MPI / torch.distributed launchers) passes its
 * own ncclComm_t to mcrt_reduce.
 * mcrt_add_accumulator: accumulator += d_src (width*height*3 doubles in
 * memory this GPU can read: another context of the same GPU, or a peer
 * buffer) - the reduction step where no communicator exists, e.g. several
 * contexts on one GPU.  Runs on this context's stream; the caller orders it
 * after the producer of d_src (mcrt_synchronize on that context). */
int mcrt_reduce(mcrt_ctx* ctx, void* nccl_comm, int root);
int mcrt_comm_init_all(int n, const int* devices, void** out_comms);
int mcrt_comm_destroy(void* nccl_comm);
int mcrt_add_accumulator(mcrt_ctx* ctx, const double* d_src);

/* The hot loop: TracePixels/RayTraceNextPixel, source/main.cpp:100-184, for
 * every pixel and for samples [sample_begin, sample_begin+sample_count):
 * PixelBuffer::Accumulate (source/core/pixelbuffer.cpp:39-46) into the device
 * accumulators.  Asynchronous on the context's stream.  Non-finite samples
 * are dropped (stated deviation). */
int mcrt_render(mcrt_ctx* ctx, uint32_t sample_begin, uint32_t sample_count,
                uint32_t max_depth, uint64_t seed);

/* Summation order of one mcrt_render pass.  The reference adds a pixel's
 * samples one by one in sample order (PixelBuffer::Accumulate,
 * source/core/pixelbuffer.cpp:39-46).  A pass does the same unless the image
 * has too few pixels (under 2^22) to keep the GPU evenly loaded and the pass
 * has at least 16 samples: then each pixel's sample range is cut
 * into L contiguous slices (slice l = samples [l*per, (l+1)*per), per =
 * sample_count / L, the last slice takes the remainder) that run on different
 * lanes; each slice is summed in sample order, and the pass adds
 * ((slice0 + slice1) + slice2) + ... to the accumulator.  L is a pure function
 * of (width, height, sample_count) - this one - so results never depend on
 * scheduling or on the number of GPUs.  mcrt_set_lanes_per_pixel forces L
 * (1 = exactly the reference's order; 0 = automatic, the default). */
uint32_t mcrt_lanes_per_pixel(uint32_t width, uint32_t height, uint32_t sample_count);
int mcrt_set_lanes_per_pixel(mcrt_ctx* ctx, uint32_t lanes);

/* Milliseconds the last mcrt_render launch took on the device, from HIP
 * events recorded on the context's stream around the kernel (synchronises). */
int mcrt_last_render_ms(mcrt_ctx* ctx, float* out_ms);

/* Mean, tonemap, gamma, clamp, x255 truncate, BGRA8: source/main.cpp:142-164
 * and GLImageBuffer::SetPixel, source/opengl/data.cpp:28-55.  spp_total is
 * the divisor (PixelBuffer::GetRayCount).  Either output may be NULL;
 * out_bgra8 = width*height*4 bytes, out_mean = width*height*3 doubles (host). */
int mcrt_resolve(mcrt_ctx* ctx, int tonemap, double gamma, double exposure,
                 uint64_t spp_total, uint8_t* out_bgra8, double* out_mean);

/* Copy the raw accumulator sums to the host (width*height*3 doubles). */
int mcrt_read_accumulator(mcrt_ctx* ctx, double* out_sums);

/* Load accumulator sums from the host (width*height*3 doubles): the inverse of
 * mcrt_read_accumulator, for resuming a render from a checkpoint.  The
 * reference keeps PixelBuffer::data in host memory for the whole run and has no
 * save/restore (source/core/pixelbuffer.h); a checkpoint is these doubles plus
 * the number of samples already rendered and the seed. */
int mcrt_write_accumulator(mcrt_ctx* ctx, const double* sums);

/* Block until everything queued on the context's stream has finished. */
int mcrt_synchronize(mcrt_ctx* ctx);

/* ------------------------------------------------------------ introspection */

typedef struct mcrt_octree_info {
    uint32_t node_count;
    uint32_t prim_ref_count;   /* entries in the per-node primitive lists   */
    uint32_t prim_count;       /* triangles + spheres                       */
    uint32_t max_depth;
    uint32_t blob_bytes;       /* size of the packed scene image            */
    uint32_t lds_resident;     /* 1 when kernels stage the whole image in LDS */
    float root_cell_min[3];    /* the root's cell: every node's cell follows from it by halving  */
    float root_cell_max[3];    /* at the centre, centre = min + 0.5f * (max - min), octant bit a */
    float pad;                 /* set = upper half on axis a.  pad: the builder's padding of cells and */
                               /* boxes; the walk routes rays with HALF of it (margin for its own rounding) */
    float far_origin_limit;    /* see mcrt_intersect */
} mcrt_octree_info;

int mcrt_octree_info_get(mcrt_ctx* ctx, mcrt_octree_info* out);

/* Download the flat octree (test infrastructure reads it to replay the same
 * traversal on the CPU with visit counters).  Arrays may be NULL to skip.
 * node_box: node_count*6 floats (min xyz, max xyz); node_words:
 * node_count*2 uint32 (child word, first primitive ref); prim_refs:
 * prim_ref_count uint32. */
int mcrt_octree_download(mcrt_ctx* ctx, float* node_box, uint32_t* node_words,
                         uint32_t* prim_refs);

/* Host-only: flatten a scene (validation, light cache, octree build, packing)
 * without touching a GPU, for inspection and CPU-side tests.  mcrt_upload_scene
 * runs exactly this and then copies the image to the device.  Errors are
 * reported through mcrt_last_error(NULL). */
typedef struct mcrt_scene_image mcrt_scene_image;
int mcrt_scene_image_create(const mcrt_scene_desc* scene, const mcrt_accel_params* accel,
                            mcrt_scene_image** out_image);
int mcrt_scene_image_destroy(mcrt_scene_image* image);
int mcrt_scene_image_info(const mcrt_scene_image* image, mcrt_octree_info* out);
int mcrt_scene_image_octree(const mcrt_scene_image* image, float* node_box, uint32_t* node_words,
                            uint32_t* prim_refs);

/* Work counters of the last mcrt_render / mcrt_intersect when counting was
 * enabled with mcrt_set_counting(ctx, 1): [0] rays, [1] nodes visited,
 * [2] triangles tested, [3] spheres tested, [4] mesh gates evaluated,
 * [5] camera samples, [6] TraceRay calls, [7] refractive splits. */
int mcrt_set_counting(mcrt_ctx* ctx, int enabled);
int mcrt_counters(mcrt_ctx* ctx, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* MCRT_H */
