/* minimal.c - the C ABI from plain C (C99): build a two-wall scene by hand, render, resolve, write a PPM.
 *
 *   gcc -std=c99 -Iinclude examples/minimal.c -Lmonte-carlo-ray-tracer_amd -lmcrt_hip -Wl,-rpath,$PWD/monte-carlo-ray-tracer_amd -lm -o minimal
 *
 * This is what a binding for the reference renderer boils down to (option A): fill
 * mcrt_object / mcrt_triangle from the scene's objects, then replace the worker threads of
 * source/main.cpp:250-262 with mcrt_render + mcrt_resolve. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mcrt.h"

static void tri(mcrt_triangle* t, const float* a, const float* b, const float* c)
{
    float e1[3], e2[3], n[3], len;
    int k;
    for (k = 0; k < 3; ++k) { t->v0[k] = a[k]; t->v1[k] = b[k]; t->v2[k] = c[k]; e1[k] = b[k] - a[k]; e2[k] = c[k] - a[k]; }
    n[0] = e1[1] * e2[2] - e2[1] * e1[2]; n[1] = e1[2] * e2[0] - e2[2] * e1[0]; n[2] = e1[0] * e2[1] - e2[0] * e1[1];   /* Triangle ctor, source/core/triangle.h:17-23 */
    len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (k = 0; k < 3; ++k) t->normal[k] = n[k] / len;
}

static void quad(mcrt_triangle* t, const float* p1, const float* p2, const float* p3, const float* p4)
{
    tri(t, p1, p2, p3);          /* TriangleMesh::AddQuad, source/objects/mesh.cpp:49-54 */
    tri(t + 1, p3, p4, p1);
}

static void bounds(mcrt_object* o, const mcrt_triangle* t, unsigned n)
{
    unsigned i; int k;
    for (k = 0; k < 3; ++k) { o->aabb_min[k] = 1e30f; o->aabb_max[k] = -1e30f; }
    for (i = 0; i < n; ++i)
        for (k = 0; k < 3; ++k) {
            const float v[3] = { t[i].v0[k], t[i].v1[k], t[i].v2[k] };
            int j;
            for (j = 0; j < 3; ++j) { if (v[j] < o->aabb_min[k]) o->aabb_min[k] = v[j]; if (v[j] > o->aabb_max[k]) o->aabb_max[k] = v[j]; }
        }
}

int main(int argc, char** argv)
{
    const unsigned w = 256, h = 192, spp = 64;
    const float f0[3] = { -5, -5, -5 }, f1[3] = { 5, -5, -5 }, f2[3] = { 5, -5, 5 }, f3[3] = { -5, -5, 5 };      /* floor, normal +y */
    const float b0[3] = { -5, -5, -5 }, b1[3] = { -5, 5, -5 }, b2[3] = { 5, 5, -5 }, b3[3] = { 5, -5, -5 };      /* back wall, normal +z */
    const float l0[3] = { -1, 4.9f, -1 }, l1[3] = { 1, 4.9f, -1 }, l2[3] = { 1, 4.9f, 1 }, l3[3] = { -1, 4.9f, 1 }; /* light, normal -y */
    mcrt_triangle tris[6];
    mcrt_object objs[3];
    mcrt_scene_desc scene;
    mcrt_camera cam;
    mcrt_ctx* ctx = NULL;
    const float eye[3] = { 0, 0, 5 }, target[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
    unsigned char* bgra;
    unsigned i;
    int rc;
    FILE* f;

    memset(objs, 0, sizeof objs);
    quad(tris + 0, f0, f3, f2, f1);
    quad(tris + 2, b0, b3, b2, b1);
    quad(tris + 4, l0, l1, l2, l3);
    for (i = 0; i < 3; ++i) {
        objs[i].kind = i == 2 ? MCRT_OBJECT_LIGHT : MCRT_OBJECT_MESH;
        objs[i].first_triangle = 2 * i; objs[i].triangle_count = 2;
        bounds(&objs[i], tris + 2 * i, 2);
        objs[i].material.color[0] = i == 0 ? 0.8 : 0.3; objs[i].material.color[1] = 0.6; objs[i].material.color[2] = i == 1 ? 0.8 : 0.3;
        objs[i].material.surface = MCRT_SURFACE_DIFFUSE; objs[i].material.diffuse = MCRT_DIFFUSE_LAMBERTIAN;
        objs[i].material.albedo = 1.0f; objs[i].material.roughness = 1.0f; objs[i].material.refractive_index = 1.52f;
    }
    objs[2].material.emission[0] = objs[2].material.emission[1] = objs[2].material.emission[2] = 10.0;
    objs[2].position[1] = 4.9f; objs[2].light_x[0] = 1.0f; objs[2].light_y[2] = 1.0f;   /* Light::SetGeometry: centre, xVector, yVector (half extents) */
    memset(&scene, 0, sizeof scene);
    scene.object_count = 3; scene.triangle_count = 6; scene.objects = objs; scene.triangles = tris;

    if ((rc = mcrt_create(0, &ctx)) != MCRT_OK) { fprintf(stderr, "mcrt_create: %s\n", mcrt_last_error(NULL)); return 1; }
    mcrt_camera_look_at(w, h, 90.0f, eye, target, up, &cam);
    if (mcrt_upload_scene(ctx, &scene, NULL) != MCRT_OK || mcrt_set_camera(ctx, &cam) != MCRT_OK || mcrt_clear(ctx) != MCRT_OK ||
        mcrt_render(ctx, 0, spp, 100, 0x5EED0001ull) != MCRT_OK) {
        fprintf(stderr, "render: %s\n", mcrt_last_error(ctx));
        return 1;
    }
    bgra = (unsigned char*)malloc((size_t)w * h * 4);
    if (mcrt_resolve(ctx, MCRT_TONEMAP_REINHARD, 2.2, 1.0, spp, bgra, NULL) != MCRT_OK) { fprintf(stderr, "resolve: %s\n", mcrt_last_error(ctx)); return 1; }
    f = fopen(argc > 1 ? argv[1] : "minimal.ppm", "wb");
    if (!f) return 1;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (i = 0; i < w * h; ++i) { fputc(bgra[4 * i + 2], f); fputc(bgra[4 * i + 1], f); fputc(bgra[4 * i], f); }
    fclose(f);
    {
        unsigned long lit = 0;
        for (i = 0; i < w * h; ++i) lit += bgra[4 * i] > 8 || bgra[4 * i + 1] > 8 || bgra[4 * i + 2] > 8;
        printf("rendered %ux%u at %u spp, %lu of %u pixels lit\n", w, h, spp, lit, w * h);
    }
    free(bgra);
    mcrt_destroy(ctx);
    return 0;
}
