"""ctypes mirror of include/mcrt.h (struct layouts and constants only).

The structs restate the reference's scene graph as POD: `Material`
(source/core/material.h:11-19), `Triangle` (source/core/triangle.h:8-23),
`Object` and subclasses (source/objects/object.h:12-33), `Camera`
(source/core/camera.h:9-58).
"""
import ctypes as C

import numpy as np

OK = 0
SURFACE_DIFFUSE, SURFACE_SPECULAR, SURFACE_DIFFUSE_SPECULAR, SURFACE_REFRACTIVE = 0, 1, 2, 3
DIFFUSE_LAMBERTIAN, DIFFUSE_OREN_NAYAR = 0, 1
OBJECT_SPHERE, OBJECT_MESH, OBJECT_LIGHT = 0, 1, 2
ACCEL_OCTREE, ACCEL_BRUTE, ACCEL_SCAN, ACCEL_AUTO, ACCEL_FLAT = 0, 1, 2, 3, 4
TONEMAP_NONE, TONEMAP_REINHARD, TONEMAP_EXPOSURE = 0, 1, 2


class Material(C.Structure):
    _fields_ = [
        ("color", C.c_double * 3),
        ("emission", C.c_double * 3),
        ("surface", C.c_int32),
        ("diffuse", C.c_int32),
        ("albedo", C.c_float),
        ("roughness", C.c_float),
        ("refractive_index", C.c_float),
        ("reserved", C.c_int32),
    ]


class Triangle(C.Structure):
    _fields_ = [
        ("v0", C.c_float * 3),
        ("v1", C.c_float * 3),
        ("v2", C.c_float * 3),
        ("normal", C.c_float * 3),
    ]


class Object(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("first_triangle", C.c_uint32),
        ("triangle_count", C.c_uint32),
        ("radius", C.c_float),
        ("position", C.c_float * 3),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("light_x", C.c_float * 3),
        ("light_y", C.c_float * 3),
        ("material", Material),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("object_count", C.c_uint32),
        ("triangle_count", C.c_uint32),
        ("objects", C.POINTER(Object)),
        ("triangles", C.POINTER(Triangle)),
        ("background", C.c_double * 3),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("fov_scale", C.c_float),
        ("view", C.c_float * 16),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
    ]


class AccelParams(C.Structure):
    _fields_ = [
        ("max_leaf_prims", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("pad_rel", C.c_float),
    ]


class OctreeInfo(C.Structure):
    _fields_ = [
        ("node_count", C.c_uint32),
        ("prim_ref_count", C.c_uint32),
        ("prim_count", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("blob_bytes", C.c_uint32),
        ("lds_resident", C.c_uint32),
        ("root_cell_min", C.c_float * 3),
        ("root_cell_max", C.c_float * 3),
        ("pad", C.c_float),
        ("far_origin_limit", C.c_float),
    ]


# numpy views of the same layouts (for fixtures and bulk construction)
MATERIAL_DTYPE = np.dtype(
    [
        ("color", "<f8", 3),
        ("emission", "<f8", 3),
        ("surface", "<i4"),
        ("diffuse", "<i4"),
        ("albedo", "<f4"),
        ("roughness", "<f4"),
        ("refractive_index", "<f4"),
        ("reserved", "<i4"),
    ],
    align=True,
)
TRIANGLE_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("normal", "<f4", 3)])
OBJECT_DTYPE = np.dtype(
    [
        ("kind", "<i4"),
        ("first_triangle", "<u4"),
        ("triangle_count", "<u4"),
        ("radius", "<f4"),
        ("position", "<f4", 3),
        ("aabb_min", "<f4", 3),
        ("aabb_max", "<f4", 3),
        ("light_x", "<f4", 3),
        ("light_y", "<f4", 3),
        ("material", MATERIAL_DTYPE),
    ],
    align=True,
)
assert MATERIAL_DTYPE.itemsize == C.sizeof(Material) == 72
assert TRIANGLE_DTYPE.itemsize == C.sizeof(Triangle) == 48
assert OBJECT_DTYPE.itemsize == C.sizeof(Object) == 152


class FlatScene:
    """Owns numpy arrays with the mcrt_object / mcrt_triangle layouts and hands
    out a SceneDesc that points into them."""

    def __init__(self, objects, triangles, background=(0.0, 0.0, 0.0)):
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE)
        self.background = tuple(float(b) for b in background)

    def desc(self):
        d = SceneDesc()
        d.object_count = len(self.objects)
        d.triangle_count = len(self.triangles)
        d.objects = self.objects.ctypes.data_as(C.POINTER(Object))
        d.triangles = self.triangles.ctypes.data_as(C.POINTER(Triangle))
        d.background[:] = self.background
        return d

    def save(self, path):
        np.savez_compressed(
            path,
            objects=self.objects.view(np.uint8),
            triangles=self.triangles.view(np.uint8),
            background=np.asarray(self.background, dtype=np.float64),
        )

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        return FlatScene(
            z["objects"].view(OBJECT_DTYPE), z["triangles"].view(TRIANGLE_DTYPE), tuple(z["background"])
        )


def camera_to_numpy(cam):
    return np.frombuffer(bytes(cam), dtype=np.uint8).copy()


def camera_from_numpy(buf):
    return Camera.from_buffer_copy(np.ascontiguousarray(buf, dtype=np.uint8).tobytes())
