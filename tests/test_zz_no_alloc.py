"""mcrt_intersect_noalloc (include/mcrt.h): the entry point that allocates no device memory - its staging lives in a
static device array of the library - against mcrt_intersect, bit for bit, on two calls in a row, the second smaller
(the array keeps the first call's bytes), plus the size limit.  Collected last (file name) so that a failure here
cannot hide the other GPU tests behind `pytest -x`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))


@pytest.mark.parametrize("name", ["cornell", "config4_2k"])
def test_noalloc_equals_allocating_entry_point(backend, pkg, scene_of, golden, name):
    backend.upload_scene(scene_of(name))
    backend.set_accel(pkg.ACCEL_AUTO)
    g = golden(f"hits_{name}.npz")
    o, d = g["origins"], g["dirs"]
    n1, n2 = len(o), 1000
    first = backend.intersect_noalloc(o, d)                       # large call: fills the staging area
    assert _same(first, backend.intersect(o, d))
    assert _same(first, (g["obj"], g["elem"], g["t"]))            # and the reference's Scene::IntersectRay
    lo = n1 // 3
    second = backend.intersect_noalloc(o[lo:lo + n2], d[lo:lo + n2])   # smaller call right after: stale bytes behind it
    assert _same(second, backend.intersect(o[lo:lo + n2], d[lo:lo + n2]))
    assert _same(second, tuple(a[lo:lo + n2] for a in first))
    assert len(backend.intersect_noalloc(o[:0], d[:0])[0]) == 0


def test_noalloc_rejects_oversize(backend, pkg, scene_of):
    backend.upload_scene(scene_of("cornell"))
    n = 400_001
    o = np.zeros((n, 3), np.float32); d = np.tile(np.array([[0, 0, -1]], np.float32), (n, 1))
    with pytest.raises(pkg.McrtError) as e:
        backend.intersect_noalloc(o, d)
    assert e.value.code == 5 and "MCRT_NOALLOC_MAX_RAYS" in str(e.value)
    obj, _, _ = backend.intersect_noalloc(o[:400_000], d[:400_000])      # exactly the maximum is served
    assert len(obj) == 400_000


def test_noalloc_two_contexts_share_the_device_array(backend, pkg, scene_of, golden):
    """A second context on the same GPU resolves the same static array; used one after the other both are right."""
    other = pkg.Backend(0)
    try:
        other.upload_scene(scene_of("hexagon"))
        backend.upload_scene(scene_of("cornell"))
        gh, gc = golden("hits_hexagon.npz"), golden("hits_cornell.npz")
        a = other.intersect_noalloc(gh["origins"][:5000], gh["dirs"][:5000])
        b = backend.intersect_noalloc(gc["origins"][:3000], gc["dirs"][:3000])
        assert _same(a, (gh["obj"][:5000], gh["elem"][:5000], gh["t"][:5000]))
        assert _same(b, (gc["obj"][:3000], gc["elem"][:3000], gc["t"][:3000]))
    finally:
        other.close()
