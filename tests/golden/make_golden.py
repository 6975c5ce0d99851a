#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (build container only).

Inputs are produced by the REFERENCE ITSELF: oracle/_ref/libicp_ref.so is the reference's own
libcpu sources (point-3d.cc, vp-tree.cc) compiled unmodified (oracle/Makefile).  Fixtures are
data only:

  data/<name>.npy            the reference's sample clouds exactly as libcpu::read_csv parses them
                             (float32 bits; data/{line,cow,horse}/*.txt of the reference repo)
  nn_<pair>_cap<c>.npz       for the first-iteration queries q = p - mean(p) against
                             VPTree(c, m - mean(m)):  mu_m, mu_p, the reference VPTree::closest
                             answer as (row in m, fp32 distance) per query, the nine
                             find_covariance sums, tree height/leaf count and a SHA-256 of the
                             reference tree's pre-order dump.  (The reference returns coordinates;
                             `row` comes from the restatement carrying indices through the same
                             build, and the script asserts (m - mu_m)[row] == reference answer
                             bit for bit before writing.)
  icp_expected.json          end-to-end results of the restated loop (oracle/icp_oracle.cc):
                             iterations, final error, transform, error trajectory.  The SVD step
                             inside is Eigen's algorithm restated (Eigen is absent offline), so
                             these are "restatement" values, cross-checked against numpy's SVD.

Run:  python3 tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import Oracle, Ref  # noqa: E402

REF_DATA = "/root/reference/data/"
CLOUDS = {
    "line_ref": "line/line_ref.txt",
    "line_translated_2_3_4": "line/line_translated_2_3_4.txt",
    "line_rotated_90_135_180": "line/line_rotated_90_135_180.txt",
    "line_scaled_2_3_4": "line/line_scaled_2_3_4.txt",
    "cow_ref": "cow/cow_ref.txt",
    "cow_tr1": "cow/cow_tr1.txt",
    "cow_tr2": "cow/cow_tr2.txt",
    "horse_ref": "horse/horse_ref.txt",
    "horse_tr1": "horse/horse_tr1.txt",
    "horse_tr2": "horse/horse_tr2.txt",
}
NN_CASES = [
    ("line_ref", "line_translated_2_3_4", (8, 32)),
    ("cow_ref", "cow_tr1", (8, 32, 256)),
    ("cow_ref", "cow_tr2", (8, 32, 256)),
    ("horse_ref", "horse_tr1", (32, 256)),
    ("horse_ref", "horse_tr2", (256,)),
]
ICP_CASES = [
    ("horse_ref", "horse_tr1", 32), ("horse_ref", "horse_tr2", 32), ("horse_ref", "horse_tr1", 256),
    ("cow_ref", "cow_tr1", 256), ("cow_ref", "cow_tr2", 256), ("cow_ref", "cow_tr1", 8),
    ("cow_ref", "cow_tr2", 8), ("line_ref", "line_translated_2_3_4", 8),
    ("line_ref", "line_translated_2_3_4", 32),
]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def tree_sha(dump):
    h = hashlib.sha256()
    for k in ("kind", "node_f", "leaf_count", "leaf_xyz"):
        h.update(np.ascontiguousarray(dump[k]).tobytes())
    return h.hexdigest()


def main():
    ref, orc = Ref(), Oracle()
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    clouds = {}
    for name, rel in CLOUDS.items():
        clouds[name] = ref.read_csv(REF_DATA + rel)
        np.save(os.path.join(HERE, "data", name + ".npy"), clouds[name])

    for mname, pname, caps in NN_CASES:
        m, p = clouds[mname], clouds[pname]
        mu_m = ref.mean(m)
        mc = ref.subtract(m, mu_m)
        mu_p = ref.mean(p)
        q = ref.subtract(p, mu_p)
        for cap in caps:
            rt = ref.tree_build(cap, mc)
            ot = orc.tree_build(cap, mc)
            rd = ref.tree_dump(rt)
            od = orc.tree_dump(ot)
            assert tree_sha(rd) == tree_sha(od)
            y_ref = ref.tree_closest(rt, q)
            d_ref = ref.tree_search_dist(rt, q)
            yo = orc.tree_closest(ot, q, counters=True)
            assert (bits(y_ref) == bits(yo["xyz"])).all() and (bits(d_ref) == bits(yo["dist"])).all()
            assert (bits(mc[yo["row"]]) == bits(y_ref)).all()
            cov = ref.find_covariance(q, y_ref)
            st = orc.tree_stats(ot)
            np.savez_compressed(
                os.path.join(HERE, f"nn_{pname}_cap{cap}.npz"),
                mu_m=mu_m, mu_p=mu_p, row=yo["row"], dist=d_ref, cov=cov,
                n_int=yo["n_int"].astype(np.uint16), n_pts=yo["n_pts"],
                height=np.uint32(st["height"]), leaves=np.uint32(st["leaves"]),
                internal=np.uint32(st["internal"]), tree_sha256=np.array(tree_sha(rd)))
            print("nn", pname, cap, st)

    expected = {}
    for mname, pname, cap in ICP_CASES:
        res = orc.icp(clouds[mname], clouds[pname], 200, 1e-5, cap)
        expected[f"{pname}_cap{cap}"] = dict(
            m=mname, p=pname, capacity=cap, iterations=int(res["iterations"]),
            final_error=float(res["errors"][-1]),
            errors=[float(e) for e in res["errors"]],
            T=[float(v) for v in res["T"].reshape(-1)],
            T_bits=[int(v) for v in bits(res["T"]).reshape(-1)],
            max_abs_newp_minus_m=float(np.abs(res["new_p"] - clouds[mname]).max()))
        print("icp", pname, cap, res["iterations"], res["errors"][-1])
    with open(os.path.join(HERE, "icp_expected.json"), "w") as f:
        json.dump(expected, f, indent=1)


if __name__ == "__main__":
    main()
