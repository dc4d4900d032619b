"""ctypes front-end of the CPU oracle (oracle/libgdo.so) and, where it has been built, of the reference itself
(oracle/_ref/libgdiet_*.so).  TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg,
__graft_entry__.smoke() and the pin/fixture scripts under oracle/ -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

NEG_INF = -0x40000000
EZ_AVX512_SC = 0x10000
EZ_SCORE_ONLY, EZ_RIGHT, EZ_GENERIC_SC, EZ_APPROX_MAX, EZ_APPROX_DROP, EZ_EXTZ_ONLY, EZ_REV_CIGAR = 1, 2, 4, 8, 0x10, 0x40, 0x80
EZ_SPLICE_FOR, EZ_SPLICE_REV, EZ_SPLICE_FLANK = 0x100, 0x200, 0x400


class GdoExtz(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_int), ("max_q", C.c_int), ("max_t", C.c_int),
                ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int),
                ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int), ("cigar", C.POINTER(C.c_uint32))]


class RefExtz(C.Structure):  # ksw_extz_t of the reference (ksw2.h:31-40): max:31|zdropped:1 share one word
    _fields_ = [("max_zd", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int),
                ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int),
                ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int), ("cigar", C.POINTER(C.c_uint32))]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load_oracle():
    path = os.path.join(HERE, "libgdo.so")
    if not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    u8p, i8p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8)
    lib.gdo_ksw_extd2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GdoExtz)]
    lib.gdo_ksw_extz2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GdoExtz)]
    lib.gdo_exact_match.argtypes = [C.c_int, u8p, C.c_int, u8p]
    lib.gdo_exact_match.restype = C.c_int
    u64p = C.POINTER(C.c_uint64)
    lib.gdo_lchain_dp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, u64p,
                                  C.POINTER(C.c_int), C.POINTER(u64p), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.gdo_lchain_dp.restype = u64p
    lib.gdo_radix_sort_128x.argtypes = [u64p, C.c_int64]
    lib.gdo_ksw_exts2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int8, C.c_int, u8p, C.POINTER(GdoExtz)]
    return lib


def have_ref(variant="lr_avx"):
    return os.path.exists(os.path.join(HERE, "_ref", "libgdiet_%s.so" % variant))


def load_ref(variant="lr_avx"):
    lib = C.CDLL(os.path.join(HERE, "_ref", "libgdiet_%s.so" % variant))
    u8p, i8p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8)
    d2 = [C.c_void_p, C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
          C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(RefExtz)]
    lib.ksw_extd2_sse.argtypes = d2
    if hasattr(lib, "ksw_extd2_avx512"):
        lib.ksw_extd2_avx512.argtypes = d2
    lib.ksw_extz2_sse.argtypes = [C.c_void_p, C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(RefExtz)]
    lib.ksw_exts2_sse.argtypes = [C.c_void_p, C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int8, C.c_int, u8p, C.POINTER(RefExtz)]
    if hasattr(lib, "mg_lchain_dp"):
        u64p = C.POINTER(C.c_uint64)
        lib.mg_lchain_dp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                     C.POINTER(C.c_int), C.POINTER(u64p), C.c_void_p]
        lib.mg_lchain_dp.restype = u64p
    lib.exact_match_sse.argtypes = [C.c_void_p, C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(RefExtz), C.POINTER(C.c_bool),
                                    C.POINTER(C.c_int)]
    return lib


def score_matrix(a, b, m=5):
    """mat[25] as built at every call site of the live path (LR/map.c:1736-1740)."""
    bb = b if b < 0 else -b
    mat = np.full((m, m), bb, dtype=np.int8)
    for i in range(m - 1):
        mat[i, i] = a
    mat[m - 1, :] = 0
    mat[:, m - 1] = 0
    return np.ascontiguousarray(mat.reshape(-1))


def _p(arr, ty):
    return arr.ctypes.data_as(C.POINTER(ty))


def _result(ez, zdropped, mx):
    n = ez.n_cigar
    cig = np.ctypeslib.as_array(ez.cigar, shape=(n,)).copy() if n > 0 else np.zeros(0, np.uint32)
    if ez.cigar:
        _libc.free(C.cast(ez.cigar, C.c_void_p))
    return dict(score=ez.score, cigar=cig.astype(np.uint32), zdropped=int(zdropped), max=int(mx), max_q=ez.max_q,
                max_t=ez.max_t, mqe=ez.mqe, mqe_t=ez.mqe_t, mte=ez.mte, mte_q=ez.mte_q, reach_end=ez.reach_end)


def oracle_extd2(lib, query, target, mat, q, e, q2, e2, w, zdrop=-1, end_bonus=0, flag=EZ_APPROX_MAX, m=5):
    ez = GdoExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    lib.gdo_ksw_extd2(len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), m, _p(mat, C.c_int8),
                      q, e, q2, e2, w, zdrop, end_bonus, flag, C.byref(ez))
    return _result(ez, ez.zdropped, ez.max)


def oracle_extz2(lib, query, target, mat, q, e, w, zdrop=-1, end_bonus=0, flag=EZ_APPROX_MAX, m=5):
    ez = GdoExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    lib.gdo_ksw_extz2(len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), m, _p(mat, C.c_int8),
                      q, e, w, zdrop, end_bonus, flag, C.byref(ez))
    return _result(ez, ez.zdropped, ez.max)


def oracle_exts2(lib, query, target, mat, q, e, q2, noncan, zdrop=-1, junc_bonus=0, flag=0, junc=None, m=5):
    """ksw_exts2 (splice-aware extension), oracle form"""
    ez = GdoExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    jp = _p(np.ascontiguousarray(junc, np.uint8), C.c_uint8) if junc is not None else None
    lib.gdo_ksw_exts2(len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), m, _p(mat, C.c_int8),
                      q, e, q2, noncan, zdrop, junc_bonus, flag, jp, C.byref(ez))
    return _result(ez, ez.zdropped, ez.max)


def ref_exts2(lib, query, target, mat, q, e, q2, noncan, zdrop=-1, junc_bonus=0, flag=0, junc=None, m=5):
    ez = RefExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    jp = _p(np.ascontiguousarray(junc, np.uint8), C.c_uint8) if junc is not None else None
    lib.ksw_exts2_sse(None, len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), m, _p(mat, C.c_int8),
                      q, e, q2, noncan, zdrop, junc_bonus, flag, jp, C.byref(ez))
    return _result(ez, ez.max_zd >> 31, ez.max_zd & 0x7fffffff)


def oracle_exact_match(lib, query, target):
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    return lib.gdo_exact_match(len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8))


def ref_extd2(lib, query, target, mat, q, e, q2, e2, w, zdrop=-1, end_bonus=0, flag=EZ_APPROX_MAX, m=5, fn="ksw_extd2_sse"):
    ez = RefExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    getattr(lib, fn)(None, len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), m, _p(mat, C.c_int8),
                     q, e, q2, e2, w, zdrop, end_bonus, flag, C.byref(ez))
    return _result(ez, ez.max_zd >> 31, ez.max_zd & 0x7fffffff)


def ref_extz2(lib, query, target, mat, q, e, w, zdrop=-1, end_bonus=0, flag=EZ_APPROX_MAX, m=5):
    ez = RefExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    lib.ksw_extz2_sse(None, len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), m, _p(mat, C.c_int8),
                      q, e, w, zdrop, end_bonus, flag, C.byref(ez))
    return _result(ez, ez.max_zd >> 31, ez.max_zd & 0x7fffffff)


def ref_exact_match(lib, query, target, mat):
    ez = RefExtz()
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    ok, mm = C.c_bool(False), C.c_int(0)
    lib.exact_match_sse(None, len(query), _p(query, C.c_uint8), len(target), _p(target, C.c_uint8), 5, _p(mat, C.c_int8),
                        0, 0, 0, 0, 0, 0, C.byref(ez), C.byref(ok), C.byref(mm))
    return int(ok.value)


def same(a, b, keys=("score", "zdropped")):
    return all(a[k] == b[k] for k in keys) and len(a["cigar"]) == len(b["cigar"]) and bool(np.all(a["cigar"] == b["cigar"]))


# ---------------------------------------------------------------------------------------------------------------
# deterministic synthetic alignment pairs (SURVEY 8d "ksw pairs" + band-edge stress); numpy Generator, fixed seeds
# ---------------------------------------------------------------------------------------------------------------
PRESETS = {  # a, b, q, e, q2, e2  (options.c:134,106,45 of the reference)
    "sr": (2, 8, 12, 2, 24, 1),
    "hifi": (1, 4, 6, 2, 26, 1),
    "ont": (2, 4, 4, 2, 24, 1),
}


def mutate(rng, seq, sub, ins, dele, n_frac=0.0):
    out = []
    for c in seq:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + ins:
            out.append(rng.integers(0, 4))
        if rng.random() < sub:
            c = (c + rng.integers(1, 4)) & 3
        if n_frac and rng.random() < n_frac:
            c = 4
        out.append(c)
    return np.array(out, dtype=np.uint8)


def make_pair(rng, tlen, sub=0.01, ins=0.003, dele=0.003, n_frac=0.0, big_indel=0, trim=0):
    target = rng.integers(0, 4, size=tlen, dtype=np.uint8)
    query = mutate(rng, target, sub, ins, dele, n_frac)
    if big_indel > 0 and len(query) > 2 * big_indel + 4:  # one long insertion in the query
        pos = int(rng.integers(1, len(query) - 1))
        query = np.concatenate([query[:pos], rng.integers(0, 4, size=big_indel, dtype=np.uint8), query[pos:]])
    elif big_indel < 0 and len(query) > -2 * big_indel + 4:  # one long deletion from the query
        pos = int(rng.integers(1, len(query) + big_indel - 1))
        query = np.concatenate([query[:pos], query[pos - big_indel:]])
    if n_frac:
        target = target.copy()
        target[rng.random(tlen) < n_frac] = 4
    if trim:
        query = query[: max(1, len(query) - trim)]
    if len(query) == 0:
        query = np.zeros(1, np.uint8)
    return np.ascontiguousarray(query), np.ascontiguousarray(target)


# ---------------------------------------------------------------------------------------------------------------
# SURVEY 8f rank 4, chaining half: mg_lchain_dp (SR/lchain.c:124) -- oracle, reference and seeded inputs
# ---------------------------------------------------------------------------------------------------------------
LCHAIN_FIELDS = ("max_dist_x", "max_dist_y", "bw", "max_skip", "max_iter", "min_cnt", "min_sc", "chn_pen_gap", "chn_pen_skip", "is_cdna", "n_seg")
_libc.malloc.argtypes = [C.c_size_t]
_libc.malloc.restype = C.c_void_p


def _lchain_out(b, n_u, u):
    nu = n_u.value
    if not b or nu == 0:
        if b:
            _libc.free(C.cast(b, C.c_void_p))
        return dict(u=np.zeros(0, np.uint64), a=np.zeros((0, 2), np.uint64))
    uu = np.ctypeslib.as_array(u, shape=(nu,)).copy()
    n_v = int((uu & 0xffffffff).sum())
    aa = np.ctypeslib.as_array(b, shape=(2 * n_v,)).copy().reshape(n_v, 2)
    _libc.free(C.cast(b, C.c_void_p)), _libc.free(C.cast(u, C.c_void_p))
    return dict(u=uu, a=aa)


def oracle_lchain(lib, a, par, want_fp=False):
    """a: uint64[n, 2] anchors (x, y) sorted by x; par: dict with LCHAIN_FIELDS"""
    a = np.ascontiguousarray(a, np.uint64)
    n = len(a)
    n_u, u = C.c_int(0), C.POINTER(C.c_uint64)()
    f, p = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int64)
    b = lib.gdo_lchain_dp(*[par[k] for k in LCHAIN_FIELDS], n, _p(a.reshape(-1), C.c_uint64), C.byref(n_u), C.byref(u), _p(f, C.c_int32), _p(p, C.c_int64))
    out = _lchain_out(b, n_u, u)
    if want_fp:
        out["f"], out["p"] = f[:n], p[:n]
    return out


def ref_lchain(lib, a, par):
    """the reference's own mg_lchain_dp: its input array is kfree'd inside (km = NULL: libc), so it gets a malloc'd copy"""
    a = np.ascontiguousarray(a, np.uint64)
    n = len(a)
    buf = _libc.malloc(max(16 * n, 16))
    C.memmove(buf, a.ctypes.data, 16 * n)
    n_u, u = C.c_int(0), C.POINTER(C.c_uint64)()
    b = lib.mg_lchain_dp(*[par[k] for k in LCHAIN_FIELDS], n, buf if n else None, C.byref(n_u), C.byref(u), None)
    if n == 0:
        _libc.free(buf)
    return _lchain_out(b, n_u, u)


def same_lchain(o, r):
    return len(o["u"]) == len(r["u"]) and bool(np.all(o["u"] == r["u"])) and o["a"].shape == r["a"].shape and bool(np.all(o["a"] == r["a"]))


def lchain_cases(rng, n_cases):
    """yield (anchors uint64[n, 2] sorted by x, parameter dict): a few colinear runs per target / strand with indel drift, repeats
    (anchors sharing query or target positions), noise anchors, sometimes two query segments (paired-end style) or cDNA-like jumps.
    Two cases in three carry minimap2's own anchor layout x = rev<<63 | tid<<32 | tpos (SR/hit.c:26: r->rev = a[k].x>>63), where the
    first anchor of the reverse strand lies >= 2^63 above the last forward one -- the unsigned distance test of SR/lchain.c:165,177
    then forces a rescan; the others keep tid<<33 | strand<<32 | tpos, where consecutive groups differ by 2^32"""
    for ci in range(n_cases):
        mm2 = ci % 3 != 0

        def X(rid, strand, tpos):
            return (strand << 63 | rid << 32 | (tpos & 0x7fffffff)) if mm2 else (rid << 33 | strand << 32 | (tpos & 0x7fffffff))
        k = int(rng.choice([11, 15, 19]))
        n_seg = 2 if ci % 9 == 4 else 1
        is_cdna = int(ci % 7 == 3)
        xs, ys = [], []
        for rid in range(int(rng.integers(1, 4))):
            for strand in (0, 1):
                for _ in range(int(rng.integers(0, 4))):
                    m = int(rng.integers(2, 400 if ci % 10 else 1500))
                    tpos, qpos = int(rng.integers(100, 200000)), int(rng.integers(0, 3000))
                    seg = int(rng.integers(0, n_seg))
                    for _ in range(m):
                        step = int(rng.integers(1, 60))
                        tpos += step + (int(rng.integers(-8, 9)) if rng.random() < 0.2 else 0) + (int(rng.integers(200, 4000)) if is_cdna and rng.random() < 0.02 else 0)
                        qpos += step
                        if rng.random() < 0.03 and n_seg > 1:
                            seg ^= 1
                        xs.append(X(rid, strand, tpos))
                        ys.append(seg << 48 | k << 32 | (qpos & 0x7fffffff))
                        if rng.random() < 0.05:  # a repeat: same query position elsewhere on the target
                            xs.append(X(rid, strand, tpos + int(rng.integers(1, 500))))
                            ys.append(seg << 48 | k << 32 | (qpos & 0x7fffffff))
        for _ in range(int(rng.integers(0, 40))):  # noise
            xs.append(X(int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 200000))))
            ys.append(int(rng.integers(0, n_seg)) << 48 | k << 32 | int(rng.integers(0, 5000)))
        a = np.array(list(zip(xs, ys)), np.uint64).reshape(-1, 2)
        if len(a):
            a = a[np.argsort(a[:, 0], kind="stable")]
        par = dict(max_dist_x=int(rng.choice([500, 5000])), max_dist_y=int(rng.choice([500, 5000])), bw=int(rng.choice([100, 500, 2000])),
                   max_skip=int(rng.choice([5, 25])), max_iter=int(rng.choice([50, 5000])), min_cnt=int(rng.choice([2, 3])),
                   min_sc=int(rng.choice([20, 40])), chn_pen_gap=float(rng.choice([0.12, 0.8])) * 0.01 * k, chn_pen_skip=float(rng.choice([0.0, 0.5])) * 0.01 * k,
                   is_cdna=is_cdna, n_seg=n_seg)
        yield a, par


def save_lchain_golden(path, gold):
    """gold: list of ((anchors, par), reference result)"""
    def pack(arrs, dt):
        offs = np.zeros(len(arrs) + 1, np.int64)
        offs[1:] = np.cumsum([len(x) for x in arrs])
        return (np.concatenate(arrs).astype(dt) if arrs and offs[-1] else np.zeros(0, dt)), offs
    ax, ao = pack([g[0][0][:, 0] for g in gold], np.uint64)
    ay, _ = pack([g[0][0][:, 1] for g in gold], np.uint64)
    ipar = np.array([[g[0][1][k] for k in LCHAIN_FIELDS if k not in ("chn_pen_gap", "chn_pen_skip")] for g in gold], np.int32)
    fpar = np.array([[g[0][1]["chn_pen_gap"], g[0][1]["chn_pen_skip"]] for g in gold], np.float32)
    u, uo = pack([g[1]["u"] for g in gold], np.uint64)
    bx, bo = pack([g[1]["a"][:, 0] for g in gold], np.uint64)
    by, _ = pack([g[1]["a"][:, 1] for g in gold], np.uint64)
    np.savez_compressed(path, ax=ax, ay=ay, ao=ao, ipar=ipar, fpar=fpar, u=u, uo=uo, bx=bx, by=by, bo=bo)
