#!/usr/bin/env python3
"""EXPERIMENT TOOL (CPU): where the steps of the octree walk go on the 1M-triangle scene.  Needs the oracle built with
-DORC_WALK_DEBUG:  gcc -std=gnu11 -O2 -fPIC -shared -ffp-contract=off -DORC_WALK_DEBUG -Iinclude -o /tmp/liborc_dbg.so oracle/mcrt_oracle.c -lm
Rays: camera rays + the diffuse bounce / shadow rays the renderer would cast from their hit points (uniform hemisphere)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import mcrt_pkg
import pyoracle as po

pkg = mcrt_pkg.load()
from mcrt_amd import scenes

n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
accel = pkg.AccelParams(*[float(x) if i == 2 else int(x) for i, x in enumerate(sys.argv[2].split(","))]) if len(sys.argv) > 2 else None
orc = po.Oracle("/tmp/liborc_dbg.so")
flat = scenes.config4(pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz")), n=n_tri, seed=42)
info, box, words, refs = pkg.build_scene_image(flat, accel)
print(f"nodes {info.node_count} refs {info.prim_ref_count} depth {info.max_depth}")
sc = orc.scene(flat)
cam = orc.camera(256, 256, 90.0, (0, 0, 5), (0, 0, 0))
rng = np.random.default_rng(3)
n = 20000
xs, ys = rng.uniform(0, 256, n).astype(np.float32), rng.uniform(0, 256, n).astype(np.float32)
o, d = orc.camera_rays(cam, xs, ys)
dbg = (C.c_uint64 * 32).in_dll(orc.lib, "orc_walk_dbg")
names = ["popped behind best", "fetched, box missed", "fetched, box hit", "interior visits", "octants pushed", "visits with a list", "leaf records"]


def run(tag, o, d):
    for i in range(32):
        dbg[i] = 0
    st = po.OctreeStats()
    obj, elem, t = orc.intersect_octree(sc, info, box, words, refs, o, d, stats=st)
    nr = len(o)
    print(f"{tag}: {nr} rays, nodes/ray {st.nodes / nr:.2f}, tris/ray {st.tris / nr:.2f}; " + ", ".join(f"{nm} {dbg[i] / nr:.2f}" for i, nm in enumerate(names)))
    return obj, elem, t


obj, elem, t = run("camera", o, d)
hit = obj >= 0
p = (o[hit] + d[hit] * t[hit, None]).astype(np.float32)
nrm = orc.surface_normal(sc, obj[hit], elem[hit], p)
flip = (nrm * d[hit]).sum(axis=1) > 0
nrm[flip] *= -1
p = (p + nrm * np.float32(20 * 1.1920929e-07)).astype(np.float32)
bd = rng.normal(size=p.shape).astype(np.float32)
bd /= np.linalg.norm(bd, axis=1, keepdims=True)
neg = (bd * nrm).sum(axis=1) < 0
bd[neg] *= -1
run("bounce", p, bd)
light = np.array([0.0, 4.999, 0.0], np.float32) + rng.uniform(-0.4, 0.4, size=p.shape).astype(np.float32) * np.array([1, 0, 1], np.float32)
ld = light - p
ld /= np.linalg.norm(ld, axis=1, keepdims=True)
run("shadow", p, ld.astype(np.float32))
