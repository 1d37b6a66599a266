#!/usr/bin/env python3
"""TEST TOOL: adversarial rays for the octree walk, CPU replay (oracle/mcrt_oracle.c:octree_hit, the mirror of
OctreeWalk::visit) against the oracle's brute force.  The ray classes are those of oracle/gen_golden.py:pad_margin_rays
(origins a few ulps either side of a split plane +- the builder's padding, one direction component tiny) at fuzz scale.

    python tools/fuzz_walk.py [--rays 1000000] [--scene config4_2k] [--gpu]

--gpu additionally runs the same rays through mcrt_intersect in every mode on GPU 0 (the oracle's brute force is the checker).
Prints one summary line per scene; exit status 1 on any mismatch."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mcrt_pkg  # noqa: E402
import pyoracle as po  # noqa: E402

pkg = mcrt_pkg.load()
import gen_golden as gg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--scene", action="append")
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--accel", default="", help="max_leaf,max_depth,pad_rel of the tree (default: the library's)")
    ap.add_argument("--soup", type=int, default=0, help="instead of the fixture scenes: the Cornell box with a soup of this many triangles (an octree that lives in HBM, so "
                    "that mcrt_intersect runs the dealt walk of the render kernel); the checker is the CPU replay of the octree, itself checked against the brute force on the first 20000 rays")
    a = ap.parse_args()
    orc = po.Oracle()
    golden = os.path.join(ROOT, "tests", "golden")
    bad_total = 0
    backend = pkg.Backend(0) if a.gpu else None
    for name in ([f"soup{a.soup}"] if a.soup else (a.scene or ["config4_2k", "cornell", "config2"])):
        if a.soup:
            from mcrt_amd import scenes
            flat = scenes.config4(pkg.FlatScene.load(os.path.join(golden, "scene_cornell.npz")), n=a.soup, seed=a.seed)
        else:
            flat = pkg.FlatScene.load(os.path.join(golden, f"scene_{name}.npz"))
        sc = orc.scene(flat)
        rng = np.random.default_rng([a.seed, len(name)])
        accel = pkg.AccelParams(*[float(x) if i == 2 else int(x) for i, x in enumerate(a.accel.split(","))]) if a.accel else None
        o, d = gg.pad_margin_rays(flat, rng, total=a.rays, levels=a.levels, accel=accel)
        fo, fd = gg.far_origin_rays(flat, rng, n=max(a.rays // 20, 10))
        o, d = np.ascontiguousarray(np.concatenate([o, fo])), np.ascontiguousarray(np.concatenate([d, fd]))
        t0 = time.time()
        info, box, words, refs = pkg.build_scene_image(flat, accel)
        got = orc.intersect_octree(sc, info, box, words, refs, o, d)
        if a.soup:
            head = orc.intersect(sc, o[:20000], d[:20000])
            want = tuple(np.concatenate([h, g[20000:]]) for h, g in zip(head, got))
        else:
            want = orc.intersect(sc, o, d)
        bad = ~((got[0] == want[0]) & (got[1] == want[1]) & (got[2].view(np.uint32) == want[2].view(np.uint32)))
        line = f"{name}: {len(o)} rays, hit fraction {float((want[0] >= 0).mean()):.3f}, cpu replay mismatches {int(bad.sum())}"
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            line += f"  first: o={o[i].tolist()} d={d[i].tolist()} want={(int(want[0][i]), int(want[1][i]), float(want[2][i]))} got={(int(got[0][i]), int(got[1][i]), float(got[2][i]))}"
        bad_total += int(bad.sum())
        if backend is not None:
            backend.upload_scene(flat, accel)
            for mode, code in (("brute", pkg.ACCEL_BRUTE), ("scan", pkg.ACCEL_SCAN), ("flat", pkg.ACCEL_FLAT), ("octree", pkg.ACCEL_OCTREE)):
                try:
                    backend.set_accel(code)
                except pkg.McrtError:
                    continue
                g = backend.intersect(o, d)
                b = ~((g[0] == want[0]) & (g[1] == want[1]) & (g[2].view(np.uint32) == want[2].view(np.uint32)))
                line += f", gpu {mode} {int(b.sum())}"
                bad_total += int(b.sum())
            backend.set_accel(pkg.ACCEL_AUTO)
        print(line + f"  ({time.time() - t0:.1f} s)", flush=True)
    if backend is not None:
        backend.close()
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
