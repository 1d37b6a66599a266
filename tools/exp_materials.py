"""Experiment (GPU box): what the Oren-Nayar BRDF, the refractive and the specular materials cost the render kernel -
the same scene with those materials replaced by Lambertian diffuse, kernel time by mcrt_last_render_ms."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcrt_pkg
pkg = mcrt_pkg.load()
name, W, H, spp = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("config2", 1024, 1024, 256)
flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
be = pkg.Backend(0)
cam = pkg.host_camera(W, H, 90.0, (0, 0, 5), (0, 0, 0))
def run(tag, f):
    be.upload_scene(f); be.set_camera(cam)
    ms = []
    for _ in range(3):
        be.clear(); be.render(0, spp, 100, 0x5EED0001); ms.append(be.last_render_ms())
    print(f"{tag:28s} {np.median(ms):8.2f} ms  {W*H*spp/np.median(ms)/1e3:8.1f} Msamples/s", flush=True)
def variant(fn):
    f = pkg.FlatScene(flat.objects.copy(), flat.triangles.copy(), flat.background); fn(f.objects["material"]); return f
run("as is", flat)
run("Oren-Nayar -> Lambert", variant(lambda m: m["diffuse"].__setitem__(slice(None), 0)))
def nospec(m): m["surface"][m["surface"] == 1] = 0
def norefr(m): m["surface"][m["surface"] == 3] = 0
run("specular -> diffuse", variant(nospec))
run("refractive -> diffuse", variant(norefr))
def alld(m): m["surface"][:] = np.where(m["surface"] > 0, 0, m["surface"]); m["diffuse"][:] = 0
run("all Lambertian diffuse", variant(alld))
