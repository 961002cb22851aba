"""Builds ABI-shaped R1CS / CRS arrays from golden groth16 cases (test helper)."""
import numpy as np

from util import R, arr, g1_aff, g2_aff, h


def csr(rows_lc):
    rp, col, val = [0], [], []
    for lc in rows_lc:
        for i, c in lc:
            col.append(int(i)); val.append(h(c))
        rp.append(len(col))
    v = arr(val, R) if val else np.zeros((0, 4), np.uint64)
    return np.array(rp, np.uint32), np.array(col, np.uint32), v


def golden_case_arrays(case):
    A = csr([r[0] for r in case["rows"]]); B = csr([r[1] for r in case["rows"]]); C = csr([r[2] for r in case["rows"]])
    crs = case["crs"]
    pts = dict(
        alpha_g1=g1_aff(crs["alpha_g1"]), beta_g1=g1_aff(crs["beta_g1"]), delta_g1=g1_aff(crs["delta_g1"]),
        beta_g2=g2_aff(crs["beta_g2"]), delta_g2=g2_aff(crs["delta_g2"]),
        A_query=np.array([g1_aff(p) for p in crs["A"]], np.uint64), B_g1=np.array([g1_aff(p) for p in crs["B1"]], np.uint64),
        B_g2=np.array([g2_aff(p) for p in crs["B2"]], np.uint64), H_query=np.array([g1_aff(p) for p in crs["H"]], np.uint64),
        L_query=np.array([g1_aff(p) for p in crs["L"]], np.uint64), m=case["m"])
    w = arr([h(x) for x in case["witness"]], R)
    r = arr([h(case["r"])], R)[0]; s = arr([h(case["s"])], R)[0]
    return A, B, C, pts, w, r, s
