"""Ad-hoc GPU diagnostics (not collected by pytest): where do GPU and golden z values differ?"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oraclelib as O  # noqa: E402
from raytracer_amd import lib  # noqa: E402

ctx = lib.Context(0)
for name in ["kat_trace_a", "kat_trace_nested", "kat_trace_mesh"]:
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    hits = ctx.trace(g["kat_rays"])
    zb, rb = hits["z"].view(np.uint32), g["kat_hit"][:, 0].copy().view(np.uint32)
    bad = np.nonzero(zb != rb)[0]
    print(name, "ids equal:", np.array_equal(hits["node"], g["kat_id"]), "z mismatches:", len(bad), "of", len(zb))
    prim = g["scene_node_prim"]
    for i in bad[:12]:
        nid = g["kat_id"][i]
        print("  ray", i, "node", nid, "prim", prim[nid] if nid >= 0 else "light", "gpu", hits["z"][i], "ref",
              g["kat_hit"][i, 0], "ulps", int(zb[i]) - int(rb[i]), "ray", g["kat_rays"][i])
    ids, cnt = np.unique(g["kat_id"][bad], return_counts=True)
    print("  mismatches by node:", dict(zip(ids.tolist(), cnt.tolist())))
    pb = (hits["p"].view(np.uint32) != g["kat_hit"][:, 1:4].copy().view(np.uint32)).any(axis=1) & (g["kat_id"] >= 0)
    nb = (hits["n"].view(np.uint32) != g["kat_hit"][:, 4:7].copy().view(np.uint32)).any(axis=1) & (g["kat_id"] >= 0)
    print("  p mismatches", pb.sum(), "n mismatches", nb.sum())
    occ = ctx.trace_shadow(g["kat_rays"], g["kat_tmax"])
    print("  shadow mismatches", (occ != g["kat_flags"][:, 2]).sum())
