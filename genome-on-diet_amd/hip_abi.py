"""ctypes binding of include/gdiet_hip.h.  Mirrors the reference's call shapes:

``Context.ksw_extd2_batch`` is the batched form of ``ksw_extd2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2,
e2, w, zdrop, end_bonus, flag, ez)`` (reference ksw2.h:68) + the exact-match pre-filter of map.c; argument meaning
and results (``score``, BAM-encoded ``cigar``) are the reference's.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NEG_INF = -0x40000000
EZ_APPROX_MAX = 0x08

# (a, b, q, e, q2, e2) of the three presets; reference options.c:134 (sr), :106 (map-hifi), :45 (map-ont default)
PRESET_SCORES = {"sr": (2, 8, 12, 2, 24, 1), "hifi": (1, 4, 6, 2, 26, 1), "ont": (2, 4, 4, 2, 24, 1)}


class GdietError(RuntimeError):
    pass


class KswScore(C.Structure):
    _fields_ = [("match", C.c_int8), ("mismatch", C.c_int8), ("sc_ambi", C.c_int8), ("q", C.c_int8), ("e", C.c_int8),
                ("q2", C.c_int8), ("e2", C.c_int8), ("reserved", C.c_int8), ("flag", C.c_int32)]

    @classmethod
    def from_preset(cls, name):
        a, b, q, e, q2, e2 = PRESET_SCORES[name]
        return cls(a, -b, 0, q, e, q2, e2, 0, EZ_APPROX_MAX)


def library_path():
    # GDIET_HIP_LIB: another build of the same library (A/B timing of kernel variants, tools/perf_dp.py); tests and bench.py never set it
    return os.environ.get("GDIET_HIP_LIB") or os.path.join(HERE, "libgdiet_hip.so")


_lib = None


def load_library():
    """dlopen libgdiet_hip.so.  torch (if it is going to be used in this process) must be imported BEFORE this so
    that both share one HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise GdietError("%s is missing: run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950)" % path)
    lib = C.CDLL(path)
    vp, i32p, i64p, u8p, u32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    lib.gdiet_hip_init.argtypes = [C.POINTER(vp), C.c_int]
    lib.gdiet_hip_destroy.argtypes = [vp]
    lib.gdiet_hip_destroy.restype = None
    lib.gdiet_hip_strerror.argtypes = [vp]
    lib.gdiet_hip_strerror.restype = C.c_char_p
    lib.gdiet_hip_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.gdiet_hip_last_kernel_mask.argtypes = [vp]
    lib.gdiet_hip_set_kernel_mode.argtypes = [vp, C.c_int]
    lib.gdiet_hip_reserve.argtypes = [vp, C.c_size_t]
    lib.gdiet_hip_set_dp_split.argtypes = [vp, C.c_int]
    lib.gdiet_hip_ksw_workspace_bytes.argtypes = [C.c_int, i64p, i64p, i32p]
    lib.gdiet_hip_ksw_workspace_bytes.restype = C.c_size_t
    lib.gdiet_hip_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.gdiet_hip_last_dp_work.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gdiet_hip_ksw_extd2_batch.argtypes = [vp, C.c_int, u8p, i64p, u8p, i64p, i32p, i32p, C.POINTER(KswScore),
                                              i32p, i32p, u32p, i64p]
    lib.gdiet_hip_ksw_extd2_batch_dev.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(KswScore),
                                                  vp, vp, vp, vp, i64p, i64p, i32p, vp]
    lib.gdiet_hip_ksw_extz2_batch.argtypes = [vp, C.c_int, u8p, i64p, u8p, i64p, i32p, C.POINTER(KswScore), i32p, i32p, u32p, i64p]
    lib.gdiet_hip_lchain_dp_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), i64p] + [C.c_int32] * 7 + [C.c_float, C.c_float, C.c_int32, C.c_int32,
                                              i32p, i64p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gdiet_hip_ksw_exts2_batch.argtypes = [vp, C.c_int, u8p, i64p, u8p, i64p, u8p, C.POINTER(C.c_int8), C.c_int8, C.c_int8, C.c_int8, C.c_int8, C.c_int32,
                                              C.c_int8, C.c_int32, i32p, i32p, u32p, i64p]
    lib.gdiet_hip_ksw_extz2_batch_ex.argtypes = [vp, C.c_int, u8p, i64p, u8p, i64p, i32p, C.POINTER(KswScore), C.c_int32, C.c_int32, i32p, i32p, u32p, i64p]
    _lib = lib
    return lib


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def pack(seqs):
    offs = np.zeros(len(seqs) + 1, np.int64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs])
    buf = np.concatenate([np.asarray(s, np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(buf, np.uint8), offs


class Context:
    """One GPU context (``gdiet_ctx``).  Raises GdietError when no gfx950 device is usable."""

    def __init__(self, device=0):
        self.lib = load_library()
        self._h = C.c_void_p()
        rc = self.lib.gdiet_hip_init(C.byref(self._h), device)
        if rc != 0:
            raise GdietError("gdiet_hip_init(device=%d) failed with %d (no gfx950 GPU?)" % (device, rc))

    def close(self):
        if self._h:
            self.lib.gdiet_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise GdietError("gdiet_hip error %d: %s" % (rc, self.lib.gdiet_hip_strerror(self._h).decode()))

    @property
    def device_name(self):
        b = C.create_string_buffer(256)
        self.lib.gdiet_hip_device_name(self._h, b, 256)
        return b.value.decode()

    def set_kernel_mode(self, mode):
        self._check(self.lib.gdiet_hip_set_kernel_mode(self._h, mode))

    def set_dp_split(self, on):
        self._check(self.lib.gdiet_hip_set_dp_split(self._h, int(on)))

    def last_kernel_mask(self):
        return self.lib.gdiet_hip_last_kernel_mask(self._h)

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self.lib.gdiet_hip_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_dp_waves(self, n):
        self.lib.gdiet_hip_set_dp_waves.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.gdiet_hip_set_dp_waves(self._h, n))

    def last_dp_clock(self):
        """(median sclk MHz, min sclk MHz, median wavefront ms) of the 64-lane DP kernel's wavefronts (gdiet_hip_last_dp_clock)"""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.lib.gdiet_hip_last_dp_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self._check(self.lib.gdiet_hip_last_dp_clock(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def last_dp_work(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.gdiet_hip_last_dp_work(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reserve(self, nbytes):
        self._check(self.lib.gdiet_hip_reserve(self._h, nbytes))

    def workspace_bytes(self, qoff, toff, w):
        return self.lib.gdiet_hip_ksw_workspace_bytes(len(w), _ptr(qoff, C.c_int64), _ptr(toff, C.c_int64), _ptr(w, C.c_int32))

    def ksw_extd2_batch(self, queries, targets, w, score, exact_score=None):
        """queries/targets: lists of nt4 uint8 arrays; w: int or per-pair ints; score: KswScore.
        Returns (scores int32[n], list of uint32 CIGAR arrays)."""
        n = len(queries)
        qbuf, qoff = pack(queries)
        tbuf, toff = pack(targets)
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(w, np.int32), (n,)))
        caps = np.array([len(q) + len(t) for q, t in zip(queries, targets)], np.int64)
        coff = np.zeros(n + 1, np.int64)
        coff[1:] = np.cumsum(caps)
        sc = np.zeros(n, np.int32)
        nc = np.zeros(n, np.int32)
        cg = np.zeros(int(coff[-1]) + 1, np.uint32)
        ex = None if exact_score is None else np.ascontiguousarray(exact_score, np.int32)
        rc = self.lib.gdiet_hip_ksw_extd2_batch(self._h, n, _ptr(qbuf, C.c_uint8), _ptr(qoff, C.c_int64),
                                                _ptr(tbuf, C.c_uint8), _ptr(toff, C.c_int64), _ptr(w, C.c_int32),
                                                None if ex is None else _ptr(ex, C.c_int32), C.byref(score),
                                                _ptr(sc, C.c_int32), _ptr(nc, C.c_int32), _ptr(cg, C.c_uint32),
                                                _ptr(coff, C.c_int64))
        self._check(rc)
        return sc, [cg[coff[i]:coff[i] + nc[i]].copy() for i in range(n)]

    def ksw_extz2_batch(self, queries, targets, w, score):
        """single-affine form (ksw_extz2_sse, reference ksw2.h:62); score.q / score.e are the gap costs"""
        n = len(queries)
        qbuf, qoff = pack(queries)
        tbuf, toff = pack(targets)
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(w, np.int32), (n,)))
        coff = np.zeros(n + 1, np.int64)
        coff[1:] = np.cumsum([len(q) + len(t) for q, t in zip(queries, targets)])
        sc, nc, cg = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(int(coff[-1]) + 1, np.uint32)
        rc = self.lib.gdiet_hip_ksw_extz2_batch(self._h, n, _ptr(qbuf, C.c_uint8), _ptr(qoff, C.c_int64), _ptr(tbuf, C.c_uint8),
                                                _ptr(toff, C.c_int64), _ptr(w, C.c_int32), C.byref(score), _ptr(sc, C.c_int32),
                                                _ptr(nc, C.c_int32), _ptr(cg, C.c_uint32), _ptr(coff, C.c_int64))
        self._check(rc)
        return sc, [cg[coff[i]:coff[i] + nc[i]].copy() for i in range(n)]

    EXTZ_FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "reach_end")

    def ksw_extz2_batch_ex(self, queries, targets, w, score, zdrop=-1, end_bonus=0):
        """exact-maximum mode of ksw_extz2_sse (score.flag = 0 or 0x40 = KSW_EZ_EXTZ_ONLY): returns (list of dicts with the scalars of
        ksw_extz_t, list of uint32 CIGAR arrays)"""
        n = len(queries)
        qbuf, qoff = pack(queries)
        tbuf, toff = pack(targets)
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(w, np.int32), (n,)))
        coff = np.zeros(n + 1, np.int64)
        coff[1:] = np.cumsum([len(q) + len(t) for q, t in zip(queries, targets)])
        ez, nc, cg = np.zeros((n, 10), np.int32), np.zeros(n, np.int32), np.zeros(int(coff[-1]) + 1, np.uint32)
        rc = self.lib.gdiet_hip_ksw_extz2_batch_ex(self._h, n, _ptr(qbuf, C.c_uint8), _ptr(qoff, C.c_int64), _ptr(tbuf, C.c_uint8),
                                                   _ptr(toff, C.c_int64), _ptr(w, C.c_int32), C.byref(score), int(zdrop), int(end_bonus),
                                                   _ptr(ez, C.c_int32), _ptr(nc, C.c_int32), _ptr(cg, C.c_uint32), _ptr(coff, C.c_int64))
        self._check(rc)
        return [dict(zip(self.EXTZ_FIELDS, (int(v) for v in ez[i]))) for i in range(n)], [cg[coff[i]:coff[i] + nc[i]].copy() for i in range(n)]

    def lchain_dp_batch(self, anchors, par):
        """mg_lchain_dp (reference mmpriv.h:102) for a batch; anchors: list of uint64[n_i, 2] arrays (x, y) sorted by x;
        par: dict with the reference's scalar arguments.  Returns a list of (u uint64[n_u], a uint64[n_v, 2]) per read."""
        n = len(anchors)
        aoff = np.zeros(n + 1, np.int64)
        aoff[1:] = np.cumsum([len(x) for x in anchors])
        tot = int(aoff[-1])
        a = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint64).reshape(-1, 2) for x in anchors]) if tot else np.zeros((0, 2), np.uint64))
        n_u, n_v = np.zeros(n, np.int32), np.zeros(n, np.int64)
        u, a_out = np.zeros(max(tot, 1), np.uint64), np.zeros((max(tot, 1), 2), np.uint64)
        rc = self.lib.gdiet_hip_lchain_dp_batch(self._h, n, _ptr(a.reshape(-1) if tot else np.zeros(2, np.uint64), C.c_uint64), _ptr(aoff, C.c_int64),
                                                *[int(par[k]) for k in ("max_dist_x", "max_dist_y", "bw", "max_skip", "max_iter", "min_cnt", "min_sc")],
                                                float(par["chn_pen_gap"]), float(par["chn_pen_skip"]), int(par["is_cdna"]), int(par["n_seg"]),
                                                _ptr(n_u, C.c_int32), _ptr(n_v, C.c_int64), _ptr(u, C.c_uint64), _ptr(a_out.reshape(-1), C.c_uint64))
        self._check(rc)
        return [(u[aoff[i]:aoff[i] + n_u[i]].copy(), a_out[aoff[i]:aoff[i] + n_v[i]].copy()) for i in range(n)]

    def ksw_exts2_batch(self, queries, targets, mat, q, e, q2, noncan, zdrop=-1, junc_bonus=0, flag=0, juncs=None):
        """ksw_exts2_sse (splice-aware extension, reference ksw2.h:71) for a batch; mat = int8[25]; juncs: None or one
        uint8 array per target; returns (list of dicts with the scalars of ksw_extz_t, list of uint32 CIGAR arrays)"""
        n = len(queries)
        qbuf, qoff = pack(queries)
        tbuf, toff = pack(targets)
        jbuf = None
        if juncs is not None:
            jbuf, _ = pack([np.ascontiguousarray(j, np.uint8) for j in juncs])
        mat = np.ascontiguousarray(mat, np.int8)
        coff = np.zeros(n + 1, np.int64)
        coff[1:] = np.cumsum([len(q_) + len(t_) + 2 for q_, t_ in zip(queries, targets)])
        ez, nc, cg = np.zeros((n, 10), np.int32), np.zeros(n, np.int32), np.zeros(int(coff[-1]) + 1, np.uint32)
        rc = self.lib.gdiet_hip_ksw_exts2_batch(self._h, n, _ptr(qbuf, C.c_uint8), _ptr(qoff, C.c_int64), _ptr(tbuf, C.c_uint8), _ptr(toff, C.c_int64),
                                                None if jbuf is None else _ptr(jbuf, C.c_uint8), _ptr(mat, C.c_int8), int(q), int(e), int(q2), int(noncan),
                                                int(zdrop), int(junc_bonus), int(flag), _ptr(ez, C.c_int32), _ptr(nc, C.c_int32), _ptr(cg, C.c_uint32),
                                                _ptr(coff, C.c_int64))
        self._check(rc)
        return [dict(zip(self.EXTZ_FIELDS, (int(v) for v in ez[i]))) for i in range(n)], [cg[coff[i]:coff[i] + nc[i]].copy() for i in range(n)]

    def ksw_extd2_batch_dev(self, n, d_qseq, d_tseq, d_exact, score, d_score, d_ncig, d_cigar, d_cigoff,
                            h_qoff, h_toff, h_w, stream=0):
        """device-pointer form: d_* are integer device addresses (torch tensor .data_ptr())."""
        rc = self.lib.gdiet_hip_ksw_extd2_batch_dev(self._h, n, d_qseq, None, d_tseq, None, None, d_exact,
                                                    C.byref(score), d_score, d_ncig, d_cigar, d_cigoff,
                                                    _ptr(h_qoff, C.c_int64), _ptr(h_toff, C.c_int64),
                                                    _ptr(h_w, C.c_int32), stream)
        self._check(rc)
