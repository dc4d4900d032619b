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
