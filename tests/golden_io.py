"""helpers to read the committed golden vectors (tests/golden/*.npz; written by oracle/pin_ksw2.py from the
reference's own ksw_extd2_sse / ksw_extz2_sse / exact_match_sse outputs)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PRESET_NAMES = ("sr", "hifi", "ont")
SCALARS = ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "reach_end")


def load_ksw(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = []
    for i in range(len(z["params"])):
        q = z["q"][z["qo"][i]:z["qo"][i + 1]]
        t = z["t"][z["to"][i]:z["to"][i + 1]]
        preset, w, zdrop, end_bonus, flag = [int(v) for v in z["params"][i]]
        cig = z["cigar_bytes"][z["cigar_off"][i]:z["cigar_off"][i + 1]].view(np.uint32)
        sc = dict(zip(SCALARS, [int(v) for v in z["scalars"][i]]))
        out.append(dict(q=q, t=t, preset=PRESET_NAMES[preset], w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag, cigar=cig, **sc))
    return out


def load_exact():
    z = np.load(os.path.join(GOLDEN, "exact_match.npz"))
    return [(z["q"][z["qo"][i]:z["qo"][i + 1]], z["t"][z["to"][i]:z["to"][i + 1]], int(z["expect"][i])) for i in range(len(z["expect"]))]


def load_exts2():
    """tests/golden/ksw2_exts2.npz (oracle/pin_rank4.py): inputs and ksw_exts2_sse's own outputs"""
    z = np.load(os.path.join(GOLDEN, "ksw2_exts2.npz"))
    out = []
    for i in range(len(z["params"])):
        q = z["q"][z["qo"][i]:z["qo"][i + 1]]
        t = z["t"][z["to"][i]:z["to"][i + 1]]
        go, ge, go2, noncan, zdrop, junc_bonus, flag, has_junc = [int(v) for v in z["params"][i]]
        junc = z["junc"][z["to"][i]:z["to"][i + 1]] if has_junc else None
        cig = z["cigar_bytes"][z["cigar_off"][i]:z["cigar_off"][i + 1]].view(np.uint32)
        sc = dict(zip(SCALARS, [int(v) for v in z["scalars"][i]]))
        out.append(dict(q=q, t=t, mat=z["mat"][i], go=go, ge=ge, go2=go2, noncan=noncan, zdrop=zdrop, junc_bonus=junc_bonus, flag=flag, junc=junc,
                        cigar=cig, **sc))
    return out


LCHAIN_INT = ("max_dist_x", "max_dist_y", "bw", "max_skip", "max_iter", "min_cnt", "min_sc", "is_cdna", "n_seg")


def load_lchain():
    """tests/golden/lchain_dp.npz (oracle/pin_rank4.py): anchors + parameters and mg_lchain_dp's own outputs (u[], rearranged anchors)"""
    z = np.load(os.path.join(GOLDEN, "lchain_dp.npz"))
    out = []
    for i in range(len(z["ipar"])):
        a = np.stack([z["ax"][z["ao"][i]:z["ao"][i + 1]], z["ay"][z["ao"][i]:z["ao"][i + 1]]], axis=1).astype(np.uint64)
        par = dict(zip(LCHAIN_INT, [int(v) for v in z["ipar"][i]]))
        par["chn_pen_gap"], par["chn_pen_skip"] = float(z["fpar"][i][0]), float(z["fpar"][i][1])
        u = z["u"][z["uo"][i]:z["uo"][i + 1]]
        b = np.stack([z["bx"][z["bo"][i]:z["bo"][i + 1]], z["by"][z["bo"][i]:z["bo"][i + 1]]], axis=1).astype(np.uint64)
        out.append(dict(a=a, par=par, u=u, b=b))
    return out
