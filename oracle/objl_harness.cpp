/* TEST INFRASTRUCTURE: the reference's vendored OBJ reader (source/helpers/OBJ_Loader.h, "OBJL") behind a C entry point, built
 * into oracle/_ref/libmcrt_objl.so from the header WHERE IT LIES (oracle/Makefile); nothing of it is copied.  Used by
 * oracle/gen_golden.py to pin the host mirror's TriangleMesh::LoadMesh: the triangle list OBJL's parser and its
 * VertexTriangluation (OBJ_Loader.h:820-985) make of a file - what the reference's LoadMesh (source/objects/mesh.cpp:70-105)
 * would consume through `Indices` if it did not ignore them.  A library of its own: mesh.cpp already includes
 * the header, and its non-inline functions cannot be linked twice. */
#include <cstdint>
#include <string>

#include "helpers/OBJ_Loader.h"

extern "C" {

/* Triangles of the whole file in LoadedIndices order: out = 9 floats per triangle (positions of its three indices).
 * Returns the number of triangles, or -1 when the file cannot be loaded; at most `cap` triangles are written. */
int objl_load_triangles(const char* path, float* out, uint32_t cap)
{
    objl::Loader loader;
    if (!loader.LoadFile(path)) return -1;
    const size_t n = loader.LoadedIndices.size() / 3;
    for (size_t t = 0; t < n && t < cap; ++t)
        for (int k = 0; k < 3; ++k) {
            const objl::Vector3& p = loader.LoadedVertices[loader.LoadedIndices[3 * t + size_t(k)]].Position;
            out[9 * t + 3 * size_t(k)] = p.X; out[9 * t + 3 * size_t(k) + 1] = p.Y; out[9 * t + 3 * size_t(k) + 2] = p.Z;
        }
    return int(n);
}

} /* extern "C" */
