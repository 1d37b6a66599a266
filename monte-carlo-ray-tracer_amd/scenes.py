"""Scene descriptions for the BASELINE.json configurations, as flat POD
(abi.FlatScene).  Geometry helpers restate, in float32 and in the reference's
operation order, what the reference's scene-construction code does:

  triangle normal   Triangle ctor, source/core/triangle.h:17-23
  AddQuad           TriangleMesh::AddQuad, source/objects/mesh.cpp:49-53
  mesh AABB         TriangleMesh::UpdateAABB, source/objects/mesh.cpp:55-68

The default Cornell box itself comes from the host library's CornellBoxScene
builder (host/) or, in tests, from the golden dump of the reference's builder.
"""
import numpy as np

from . import abi

F = np.float32


def triangle_normals(v0, v1, v2):
    """normalize(cross(v1 - v0, v2 - v0)) with glm's operation order."""
    v0, v1, v2 = (np.asarray(a, F).reshape(-1, 3) for a in (v0, v1, v2))
    u = v1 - v0
    v = v2 - v0
    cx = u[:, 1] * v[:, 2] - v[:, 1] * u[:, 2]
    cy = u[:, 2] * v[:, 0] - v[:, 2] * u[:, 0]
    cz = u[:, 0] * v[:, 1] - v[:, 0] * u[:, 1]
    d = (cx * cx + cy * cy) + cz * cz
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = F(1.0) / np.sqrt(d, dtype=F)
        return np.stack([cx * inv, cy * inv, cz * inv], axis=1).astype(F)


def make_triangles(v0, v1, v2):
    v0, v1, v2 = (np.asarray(a, F).reshape(-1, 3) for a in (v0, v1, v2))
    t = np.zeros(len(v0), abi.TRIANGLE_DTYPE)
    t["v0"], t["v1"], t["v2"] = v0, v1, v2
    t["normal"] = triangle_normals(v0, v1, v2)
    return t


def quad(p1, p2, p3, p4):
    """AddQuad: (p1,p2,p3), (p3,p4,p1)."""
    return make_triangles([p1, p3], [p2, p4], [p3, p1])


def mesh_aabb(position, tris):
    pts = np.concatenate([np.asarray(position, F).reshape(1, 3), tris["v0"], tris["v1"], tris["v2"]], axis=0)
    return pts.min(axis=0), pts.max(axis=0)


def material(color=(1.0, 1.0, 1.0), emission=(0.0, 0.0, 0.0), surface=abi.SURFACE_DIFFUSE, diffuse=abi.DIFFUSE_LAMBERTIAN, albedo=1.0,
             roughness=1.0, ior=1.52):
    """Defaults are those of struct Material, source/core/material.h:13-19."""
    m = np.zeros((), abi.MATERIAL_DTYPE)
    m["color"] = np.broadcast_to(np.asarray(color, np.float64), 3)
    m["emission"] = np.broadcast_to(np.asarray(emission, np.float64), 3)
    m["surface"], m["diffuse"] = surface, diffuse
    m["albedo"], m["roughness"], m["refractive_index"] = albedo, roughness, ior
    return m


def add_mesh(flat, tris, position=(0.0, 0.0, 0.0), mat=None):
    """Append one TriangleMesh object (CreateObject<TriangleMesh> + triangles + UpdateAABB)."""
    tris = np.ascontiguousarray(tris, abi.TRIANGLE_DTYPE)
    o = np.zeros(1, abi.OBJECT_DTYPE)
    o["kind"] = abi.OBJECT_MESH
    o["first_triangle"] = len(flat.triangles)
    o["triangle_count"] = len(tris)
    o["position"] = np.asarray(position, F)
    mn, mx = mesh_aabb(position, tris)
    o["aabb_min"], o["aabb_max"] = mn, mx
    o["material"] = mat if mat is not None else material()
    return abi.FlatScene(np.concatenate([flat.objects, o]), np.concatenate([flat.triangles, tris]), flat.background)


# --------------------------------------------------------------------- config 2
# Geometry of the three stacked cubes of the reference's binaries/box_stack.obj
# asset, restated from its facts (24 vertices, 18 quads); faces triangulated by
# their indices the way the reference's OBJ library splits a quad - (a,b,d),
# (b,c,d), source/helpers/OBJ_Loader.h:890-928; tests/golden/obj_triangulation.npz
# holds that library's own output for these faces - not by the broken sliding
# window of TriangleMesh::LoadMesh (source/objects/mesh.cpp:92-103).
_BOX_STACK_FACES = ((2, 4, 3, 1), (4, 8, 7, 3), (8, 6, 5, 7), (6, 2, 1, 5), (1, 3, 7, 5), (6, 8, 4, 2))


def _axis_cube(h, y0, y1):
    return [(h, y0, h), (h, y1, h), (-h, y0, h), (-h, y1, h), (h, y0, -h), (h, y1, -h), (-h, y0, -h), (-h, y1, -h)]


def _box_stack_cubes():
    r = 0.754531
    return [
        _axis_cube(0.234105, 2.076805, 2.545015),
        [(0.0, 1.004430, r), (0.0, 2.071499, r), (-r, 1.004430, 0.0), (-r, 2.071499, 0.0), (r, 1.004430, -0.0), (r, 2.071499, -0.0),
         (-0.0, 1.004430, -r), (-0.0, 2.071499, -r)],
        [(-1.0, -1.0, 1.0), (-1.0, 1.0, 1.0), (-1.0, -1.0, -1.0), (-1.0, 1.0, -1.0), (1.0, -1.0, 1.0), (1.0, 1.0, 1.0), (1.0, -1.0, -1.0),
         (1.0, 1.0, -1.0)],
    ]


def box_stack_obj_text():
    """The three cubes as an OBJ file: 24 `v` records, 18 quad `f` records with 1-based indices (the asset's shape)."""
    lines = ["# three stacked cubes", "o stack"]
    faces = []
    for k, verts in enumerate(_box_stack_cubes()):
        for v in verts:
            lines.append("v %.9g %.9g %.9g" % tuple(float(np.float32(x)) for x in v))
        for f in _BOX_STACK_FACES:
            faces.append("f " + " ".join(str(8 * k + i) for i in f))
    return "\n".join(lines + faces) + "\n"


def box_stack_triangles(offset=(0.0, 0.0, 0.0)):
    cubes = _box_stack_cubes()
    off = np.asarray(offset, F)
    a, b, c = [], [], []
    for verts in cubes:
        v = np.asarray(verts, F) + off
        for f in _BOX_STACK_FACES:
            p = [v[i - 1] for i in f]
            # a quad as the reference's OBJ library splits it (source/helpers/OBJ_Loader.h:890-928): (0, 1, 3), (1, 2, 3)
            a += [p[0], p[1]]; b += [p[1], p[2]]; c += [p[3], p[3]]
    return make_triangles(a, b, c)


def config2(cornell):
    """Cornell box + the box-stack mesh (36 triangles) at (0,-4,1.5), Diffuse/Lambert 0.5."""
    pos = (0.0, -4.0, 1.5)
    return add_mesh(cornell, box_stack_triangles(pos), pos, material(color=0.5))


# --------------------------------------------------------------------- config 3
def config3(cornell):
    """Oren-Nayar + heavy refraction variant: spheres 3 and 5 and all three boxes
    Oren-Nayar (roughness 0.5); spheres 4 and 6 refractive (ior 1.52 / 1.4); box 8 refractive."""
    f = abi.FlatScene(cornell.objects.copy(), cornell.triangles.copy(), cornell.background)
    o = f.objects
    for i in (3, 5, 7, 8, 9):
        o["material"]["diffuse"][i] = abi.DIFFUSE_OREN_NAYAR
        o["material"]["roughness"][i] = 0.5
    o["material"]["surface"][4] = abi.SURFACE_REFRACTIVE
    o["material"]["refractive_index"][4] = 1.52
    o["material"]["surface"][6] = abi.SURFACE_REFRACTIVE
    o["material"]["refractive_index"][6] = 1.4
    o["material"]["surface"][8] = abi.SURFACE_REFRACTIVE
    return f


# --------------------------------------------------------------------- config 4
def pcg32_stream(n, seed=42, seq=54):
    """n outputs of PCG32 (XSH-RR 64/32), vectorised through the closed form of the LCG."""
    mult = np.uint64(6364136223846793005)
    inc = np.uint64((seq << 1) | 1)
    with np.errstate(over="ignore"):
        # pcg32_srandom: state = 0; step; state += seed; step
        s0 = (np.uint64(0) * mult + inc)
        s0 = (s0 + np.uint64(seed)) * mult + inc
        pw = np.ones(n, np.uint64)
        pw[1:] = mult
        pw = np.cumprod(pw, dtype=np.uint64)                    # mult^k, k = 0..n-1 (mod 2^64)
        geo = np.zeros(n, np.uint64)
        geo[1:] = np.cumsum(pw[:-1], dtype=np.uint64)           # sum_{j<k} mult^j
        state = pw * s0 + geo * inc                             # state before output k
        xorshifted = (((state >> np.uint64(18)) ^ state) >> np.uint64(27)).astype(np.uint32)
        rot = (state >> np.uint64(59)).astype(np.uint32)
        return (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))


def random_mesh_triangles(n, seed=42):
    """n triangles: centre uniform in [-4.5,4.5]x[-4.5,4.0]x[-4.5,4.5], vertices = centre + offsets uniform in
    [-0.05,0.05]^3; 12 PCG32 draws per triangle (centre xyz, then v0, v1, v2 offsets)."""
    u = (pcg32_stream(12 * n, seed) >> np.uint32(8)).astype(F) * F(2.0 ** -24)
    u = u.reshape(n, 12)
    lo = np.array([-4.5, -4.5, -4.5], F)
    hi = np.array([4.5, 4.0, 4.5], F)
    centre = lo + (hi - lo) * u[:, 0:3]
    offs = F(-0.05) + F(0.1) * u[:, 3:12]
    v0 = centre + offs[:, 0:3]
    v1 = centre + offs[:, 3:6]
    v2 = centre + offs[:, 6:9]
    return make_triangles(v0, v1, v2)


def config4(cornell, n=1_000_000, seed=42):
    """Cornell walls + light + one n-triangle random mesh, Diffuse/Lambert 0.5."""
    keep = [0, 1, 2, 10]
    objs = cornell.objects[keep].copy()
    tris = []
    first = 0
    for k in range(len(objs)):
        a, c = int(objs["first_triangle"][k]), int(objs["triangle_count"][k])
        tris.append(cornell.triangles[a:a + c])
        objs["first_triangle"][k] = first
        first += c
    base = abi.FlatScene(objs, np.concatenate(tris), cornell.background)
    # keep the light last (object order is the tie-break order; the mesh sits before it like the example objects do)
    mesh_first = abi.FlatScene(base.objects[:3], base.triangles[:int(base.objects["first_triangle"][3])], base.background)
    with_mesh = add_mesh(mesh_first, random_mesh_triangles(n, seed), (0.0, 0.0, 0.0), material(color=0.5))
    light = base.objects[3:4].copy()
    ltris = base.triangles[int(base.objects["first_triangle"][3]):]
    light["first_triangle"] = len(with_mesh.triangles)
    return abi.FlatScene(np.concatenate([with_mesh.objects, light]), np.concatenate([with_mesh.triangles, ltris]), base.background)


def many_objects(cornell, meshes=120, tris_per_mesh=6, seed=7):
    """The Cornell box plus `meshes` small triangle meshes of their own (clumps of the random soup, scaled up so that they
    are hit): a scene with more objects than the 64 the per-ray mesh-gate cache holds."""
    f = cornell
    tri = random_mesh_triangles(meshes * tris_per_mesh, seed)
    for m in range(meshes):
        t = tri[m * tris_per_mesh:(m + 1) * tris_per_mesh]
        centre = t["v0"][0].copy()
        shift = (F(0.1) * np.arange(tris_per_mesh, dtype=F))[:, None] * np.array([1.0, 0.5, -0.7], F)
        v = [(centre + (t[k] - t["v0"]) * F(8.0) + shift).astype(F) for k in ("v0", "v1", "v2")]
        f = add_mesh(f, make_triangles(*v), tuple(float(x) for x in centre), material(color=0.3 + 0.005 * m))
    return f


# --------------------------------------------------------------------- several lights
def more_lights(base):
    """`base` (a Cornell-box scene) with two more emitters: a sphere light of radius 0.4 and a point light (a sphere of radius 0,
    source/scene.cpp:485-492).  Exercises the light loop of source/scene.cpp:127-141 with three lights of three kinds."""
    n = len(base.objects)
    objs = np.concatenate([base.objects, base.objects[3:5]]).copy()
    for k, (pos, radius, emission) in enumerate([((-2.5, 2.0, 1.0), 0.4, (6.0, 4.0, 2.0)), ((2.0, 3.0, -1.0), 0.0, (3.0, 5.0, 8.0))]):
        o = objs[n + k:n + k + 1]
        o["kind"] = abi.OBJECT_SPHERE
        o["first_triangle"] = 0
        o["triangle_count"] = 0
        o["radius"] = radius
        o["position"] = pos
        o["aabb_min"] = np.asarray(pos, F) - F(radius)
        o["aabb_max"] = np.asarray(pos, F) + F(radius)
        o["material"]["emission"] = emission
        o["material"]["color"] = emission
    return abi.FlatScene(objs, base.triangles.copy(), base.background)


# --------------------------------------------------------------------- random scenes (parity fuzzing)
def add_sphere(flat, position, radius, mat=None):
    """Append one SphereObject (source/objects/sphere.h:8-14)."""
    o = np.zeros(1, abi.OBJECT_DTYPE)
    o["kind"] = abi.OBJECT_SPHERE
    o["radius"] = radius
    o["position"] = np.asarray(position, F)
    o["aabb_min"] = np.asarray(position, F) - F(radius)
    o["aabb_max"] = np.asarray(position, F) + F(radius)
    o["material"] = mat if mat is not None else material()
    return abi.FlatScene(np.concatenate([flat.objects, o]), flat.triangles, flat.background)


def random_scene(cornell, seed):
    """The walls and the quad light of the Cornell box around a random population: spheres and small random meshes with materials
    of every kind (Lambert / Oren-Nayar, mirror, glass), sometimes a second emitter (sphere or point).  Same generator, same scene."""
    rng = np.random.default_rng([20261111, seed])
    keep = [0, 1, 2, 10]
    objs = cornell.objects[keep].copy()
    tris, first = [], 0
    for k in range(len(objs)):
        a, c = int(objs["first_triangle"][k]), int(objs["triangle_count"][k])
        tris.append(cornell.triangles[a:a + c])
        objs["first_triangle"][k] = first
        first += c
    f = abi.FlatScene(objs, np.concatenate(tris), tuple(rng.uniform(0.0, 0.2, 3)) if seed % 2 else (0.0, 0.0, 0.0))

    def mat():
        kind = rng.integers(0, 6)
        color = rng.uniform(0.2, 1.0, 3)
        if kind == 0:
            return material(color=color, surface=abi.SURFACE_SPECULAR)
        if kind == 1:
            return material(color=color, surface=abi.SURFACE_REFRACTIVE, ior=float(rng.uniform(1.1, 1.8)))
        if kind in (2, 3):
            return material(color=color, diffuse=abi.DIFFUSE_OREN_NAYAR, roughness=float(rng.uniform(0.05, 1.5)), albedo=float(rng.uniform(0.5, 1.0)))
        return material(color=color, albedo=float(rng.uniform(0.5, 1.0)))

    for _ in range(int(rng.integers(2, 9))):
        f = add_sphere(f, rng.uniform(-3.5, 3.5, 3), float(rng.uniform(0.2, 1.4)), mat())
    for _ in range(int(rng.integers(1, 7))):
        n = int(rng.integers(1, 25))
        centre = rng.uniform(-3.5, 3.5, 3).astype(F)
        v = [(centre + rng.uniform(-1.2, 1.2, (n, 3))).astype(F) for _ in range(3)]
        f = add_mesh(f, make_triangles(*v), tuple(float(x) for x in centre), mat())
    extra = rng.integers(0, 3)
    if extra:
        f = add_sphere(f, rng.uniform(-3.0, 3.0, 3), 0.0 if extra == 1 else float(rng.uniform(0.2, 0.6)),
                       material(color=(4.0, 5.0, 6.0), emission=(4.0, 5.0, 6.0)))
    return f


def transformed(base, scale, shift):
    """`base` scaled about the origin and moved by `shift` along every axis, in float arithmetic; mesh boxes re-made from the rounded
    vertices as TriangleMesh::UpdateAABB does.  (Normals stay: a uniform scale and a translation do not turn them.)"""
    objs, tris = base.objects.copy(), base.triangles.copy()
    for k in ("v0", "v1", "v2"):
        tris[k] = (tris[k] * F(scale) + F(shift)).astype(F)
    for k in ("position", "aabb_min", "aabb_max"):
        objs[k] = (objs[k] * F(scale) + F(shift)).astype(F)
    for k in ("radius", "light_x", "light_y"):
        objs[k] = (objs[k] * F(scale)).astype(F)
    for i in range(len(objs)):
        a, c = int(objs["first_triangle"][i]), int(objs["triangle_count"][i])
        if c:
            pts = np.concatenate([tris[k][a:a + c] for k in ("v0", "v1", "v2")] + [objs["position"][i:i + 1]])
            objs["aabb_min"][i], objs["aabb_max"][i] = pts.min(axis=0), pts.max(axis=0)
    return abi.FlatScene(objs, tris, base.background)
