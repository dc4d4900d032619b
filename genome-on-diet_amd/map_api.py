"""ctypes binding of the B1 part of include/gdiet_hip.h: device index + per-read mapping batches + SAM records.

Mirrors the reference's call shape: ``Mapper(ctx, ref_names, ref_seqs, preset_options)`` stands for
``mm_idx_gen`` + ``mm_mapopt_update``; ``Mapper.map(reads)`` for one ``kt_for(worker_for)`` pass over a batch
(reference map.c, step 1 of worker_pipeline); ``Mapper.sam(...)`` for ``mm_write_sam3``.
"""
import ctypes as C

import numpy as np

from .hip_abi import GdietError, load_library

F_NO_PRINT_2ND = 0x4000
F_SR, F_FRAG_MODE = 0x1000, 0x2000


class MapOpt(C.Structure):
    _fields_ = [("flag", C.c_int64), ("a", C.c_int32), ("b", C.c_int32), ("q", C.c_int32), ("e", C.c_int32), ("q2", C.c_int32),
                ("e2", C.c_int32), ("bw", C.c_uint32), ("min_dp_max", C.c_int32), ("best_n", C.c_int32), ("q_occ_frac", C.c_float),
                ("mid_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32), ("max_frag_len", C.c_int32),
                ("vt_dis", C.c_uint32), ("vt_nb_loc", C.c_uint32), ("vt_cov", C.c_float), ("vt_f", C.c_float), ("vt_df1", C.c_float),
                ("vt_df2", C.c_float), ("max_max_gap", C.c_uint32), ("max_min_gap", C.c_uint32), ("max_seeds", C.c_float),
                ("min_cnt", C.c_float), ("rec_threshold_frac", C.c_float), ("bw_frac", C.c_float), ("bw_min", C.c_int32), ("bw_max", C.c_int32),
                ("AF_max_loc", C.c_int32)]


class Reg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("id", "cnt", "rid", "score", "qs", "qe", "rs", "re", "parent", "subsc", "mlen", "blen")] + \
               [("mapq", C.c_uint32), ("rev", C.c_uint32), ("sam_pri", C.c_uint32), ("dp_score", C.c_int32), ("dp_max", C.c_int32),
                ("n_ambi", C.c_uint32), ("n_cigar", C.c_uint32), ("cigar", C.POINTER(C.c_uint32))]


# the canonical command lines of the reference's README, as option values.
# idx: k, w, pattern; min/max_mid_occ: the clamp mm_mapopt_update applies (options.c:64-70, presets :45/:106)
PRESETS = {
    "hifi": dict(k=19, w=19, Z="10", W=2, a=1, b=4, q=6, e=2, q2=26, e2=1, bw=1000, min_dp_max=400, best_n=1, max_seeds=0.2,
                 vt_dis=650, vt_nb_loc=5, vt_df1=0.0106, vt_df2=0.2, vt_cov=0.04, vt_f=0.04, max_min_gap=4000, max_max_gap=50000,
                 occ_dist=500, min_mid_occ=50, max_mid_occ=500, flag=0),
    "ont": dict(k=15, w=10, Z="10", W=2, a=2, b=4, q=4, e=2, q2=24, e2=1, bw=1300, min_dp_max=35000, best_n=1, max_seeds=0.2,
                vt_dis=1000, vt_nb_loc=3, vt_df1=0.007, vt_df2=0.007, vt_cov=0.3, vt_f=0.04, max_min_gap=4000, max_max_gap=50000,
                occ_dist=500, min_mid_occ=10, max_mid_occ=1000000, flag=0),
    # README.md:41: -ax sr -Z 10 -W 2 -i 2 -k 21 -w 11 -N 1 -r 0.05,150,200 -n 0.95,0.3 -s 100 --AF_max_loc 2 --secondary=yes
    # (preset options.c:130-148: flag SR|FRAG_MODE|HEAP_SORT, NO_PRINT_2ND cleared by --secondary=yes; mid_occ fixed at 1000)
    "sr": dict(k=21, w=11, Z="10", W=2, a=2, b=8, q=12, e=2, q2=24, e2=1, bw=0, min_dp_max=100, best_n=1, max_seeds=2.0,
               vt_dis=0, vt_nb_loc=0, vt_df1=0.0, vt_df2=0.0, vt_cov=0.0, vt_f=0.0, max_min_gap=0, max_max_gap=0,
               occ_dist=500, mid_occ=1000, min_mid_occ=0, max_mid_occ=0, flag=F_SR | F_FRAG_MODE, max_frag_len=800,
               min_cnt=0.95, rec_threshold_frac=0.3, bw_frac=0.05, bw_min=150, bw_max=200, AF_max_loc=2),
}


def _bind(lib):
    if getattr(lib, "_map_bound", False):
        return
    vp = C.c_void_p
    cpp = C.POINTER(C.c_char_p)
    i32p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    lib.gdiet_hip_index_build.argtypes = [vp, C.POINTER(vp), C.c_int, cpp, cpp, u32p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
    u64p = C.POINTER(C.c_uint64)
    lib.gdiet_hip_index_import.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, cpp, u32p, u64p, u32p,
                                           C.c_uint64, u64p, u32p, u64p]
    lib.gdiet_hip_index_export.argtypes = [vp, u64p, u64p, u64p, u64p, u32p, u64p, u32p, u64p]
    lib.gdiet_hip_index_load_mmi.argtypes = [vp, C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_int]
    lib.gdiet_hip_index_dump_mmi.argtypes = [vp, vp, C.c_char_p, C.c_int]
    lib.gdiet_hip_index_destroy.argtypes = [vp, vp]
    lib.gdiet_hip_index_destroy.restype = None
    lib.gdiet_hip_index_cal_max_occ.argtypes = [vp, C.c_float]
    lib.gdiet_hip_index_cal_max_occ.restype = C.c_int32
    lib.gdiet_hip_index_n_keys.argtypes = [vp]
    lib.gdiet_hip_index_n_keys.restype = C.c_uint64
    lib.gdiet_hip_map_batch.argtypes = [vp, vp, C.POINTER(MapOpt), C.c_int, cpp, i32p, i32p, C.POINTER(C.POINTER(Reg))]
    lib.gdiet_hip_free_regs.argtypes = [C.c_int, i32p, C.POINTER(C.POINTER(Reg))]
    lib.gdiet_hip_free_regs.restype = None
    lib.gdiet_hip_map_batch_multi.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(MapOpt), C.c_int, cpp, i32p, i32p, C.POINTER(C.POINTER(Reg))]
    lib.gdiet_hip_read_ranges_by_cost.argtypes = [C.c_int, i32p, C.c_int, C.c_int32, i32p]
    lib.gdiet_hip_map_frag.argtypes = [vp, vp, C.POINTER(MapOpt), C.c_int, i32p, cpp, i32p, C.POINTER(C.POINTER(Reg))]
    lib.gdiet_hip_seed_batch.argtypes = [vp, vp, C.POINTER(MapOpt), C.c_int, cpp, i32p, i32p, u32p, u32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                         C.POINTER(vp), C.POINTER(vp)]
    lib.gdiet_hip_map_failed_reads.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]
    lib.gdiet_hip_batch_upload.argtypes = [vp, C.POINTER(vp), C.c_int, cpp, i32p]
    lib.gdiet_hip_map_uploaded.argtypes = [vp, vp, C.POINTER(MapOpt), vp, i32p, C.POINTER(C.POINTER(Reg))]
    lib.gdiet_hip_map_submit.argtypes = [vp, vp, C.POINTER(MapOpt), vp, i32p, C.POINTER(C.POINTER(Reg)), C.POINTER(vp)]
    lib.gdiet_hip_map_wait.argtypes = [vp, vp]
    lib.gdiet_hip_set_inflight.argtypes = [vp, C.c_int]
    lib.gdiet_hip_batch_destroy.argtypes = [vp, vp]
    lib.gdiet_hip_batch_destroy.restype = None
    lib.gdiet_hip_map_stage_seconds.argtypes = [vp, C.POINTER(C.c_double)]
    lib.gdiet_hip_set_host_threads.argtypes = [vp, C.c_int]
    lib.gdiet_hip_set_map_lanes.argtypes = [vp, C.c_int]
    lib.gdiet_hip_sam_record.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(Reg), C.c_int32, C.c_int32,
                                         C.c_int64, C.c_char_p, C.c_size_t]
    lib.gdiet_hip_sam_record.restype = C.c_size_t
    lib.gdiet_hip_sam_batch.argtypes = [vp, vp, C.c_int, cpp, cpp, cpp, i32p, i32p, C.POINTER(C.POINTER(Reg)), C.c_int64, C.POINTER(vp)]
    lib.gdiet_hip_sam_batch.restype = C.c_size_t
    lib.gdiet_hip_paf_batch.argtypes = [vp, vp, C.c_int, cpp, i32p, i32p, C.POINTER(C.POINTER(Reg)), C.c_int64, C.POINTER(vp)]
    lib.gdiet_hip_paf_batch.restype = C.c_size_t
    lib._map_bound = True


class MapResult:
    """regs of one batch; frees the C arrays when dropped"""

    def __init__(self, lib, n, n_regs, regs):
        self.lib, self.n, self.n_regs, self.regs = lib, n, n_regs, regs

    def __del__(self):
        try:
            self.lib.gdiet_hip_free_regs(self.n, self.n_regs, self.regs)
        except Exception:
            pass


def read_ranges_by_cost_c(lens, parts, band=1000):
    """gdiet_hip_read_ranges_by_cost: the split gdiet_hip_map_batch_multi applies (host arithmetic; shard.read_ranges_by_cost is its mirror)"""
    lib = load_library()
    _bind(lib)
    lens = np.ascontiguousarray(lens, np.int32)
    bounds = np.zeros(parts + 1, np.int32)
    rc = lib.gdiet_hip_read_ranges_by_cost(len(lens), lens.ctypes.data_as(C.POINTER(C.c_int32)), parts, band, bounds.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc:
        raise GdietError("gdiet_hip_read_ranges_by_cost: %d" % rc)
    return [int(b) for b in bounds]


def map_multi(mappers, reads):
    """gdiet_hip_map_batch_multi: one mini-batch fanned out over the mappers' contexts (one per GPU, index replicated), contiguous read
    ranges of equal DP cost, records gathered in input order.  The mappers must have been made with the same options."""
    m0 = mappers[0]
    n, reads, arr, lens = m0._arrays(reads)
    k = len(mappers)
    ctxs = (C.c_void_p * k)(*[m.ctx._h for m in mappers])
    idxs = (C.c_void_p * k)(*[m._idx for m in mappers])
    n_regs = (C.c_int32 * n)()
    regs = (C.POINTER(Reg) * n)()
    rc = m0.lib.gdiet_hip_map_batch_multi(k, ctxs, idxs, C.byref(m0.opt), n, arr, lens.ctypes.data_as(C.POINTER(C.c_int32)), n_regs, regs)
    m0.ctx._check(rc)
    return MapResult(m0.lib, n, n_regs, regs)


class Mapper:
    def map_frag(self, seqs):
        """gdiet_hip_map_frag: one fragment (list of segment sequences), mm_map_frag's call shape"""
        n, seqs_b, arr, lens = self._arrays(seqs)
        n_regs = (C.c_int32 * n)()
        regs = (C.POINTER(Reg) * n)()
        self.ctx._check(self.lib.gdiet_hip_map_frag(self.ctx._h, self._idx, C.byref(self.opt), n, lens.ctypes.data_as(C.POINTER(C.c_int32)), arr, n_regs, regs))
        return MapResult(self.lib, n, n_regs, regs)

    def seed_batch(self, reads):
        """gdiet_hip_seed_batch: per read (shift, tmp_extracted_len, n_mv, seeds [(n, q_pos)], occurrences [y values of the seeds, back to back])"""
        n, reads_b, arr, lens = self._arrays(reads)
        shift, tel, n_mv = np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        so, oo = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        sd, oc = C.c_void_p(), C.c_void_p()
        i32p, u32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)
        self.ctx._check(self.lib.gdiet_hip_seed_batch(self.ctx._h, self._idx, C.byref(self.opt), n, arr, lens.ctypes.data_as(i32p), shift.ctypes.data_as(i32p),
                                                      tel.ctypes.data_as(u32p), n_mv.ctypes.data_as(u32p), so.ctypes.data_as(i64p), oo.ctypes.data_as(i64p),
                                                      C.byref(sd), C.byref(oc)))
        try:
            seeds = np.ctypeslib.as_array(C.cast(sd, u32p), shape=(int(so[n]) * 2,)).reshape(-1, 2).copy() if so[n] else np.zeros((0, 2), np.uint32)
            occ = np.ctypeslib.as_array(C.cast(oc, C.POINTER(C.c_uint64)), shape=(int(oo[n]),)).copy() if oo[n] else np.zeros(0, np.uint64)
        finally:
            libc = C.CDLL(None)
            libc.free(sd), libc.free(oc)
        return [dict(shift=int(shift[i]), tel=int(tel[i]), n_mv=int(n_mv[i]), seeds=seeds[so[i]:so[i + 1]], occ=occ[oo[i]:oo[i + 1]]) for i in range(n)]

    def failed_reads(self):
        """(reads the last map call left unmapped because of a degenerate DP box, the same since the context was made, description)"""
        a, b, w = C.c_int64(), C.c_int64(), C.c_char_p()
        self.lib.gdiet_hip_map_failed_reads(self.ctx._h, C.byref(a), C.byref(b), C.byref(w))
        return a.value, b.value, (w.value or b"").decode()

    def __init__(self, ctx, names, seqs, preset="hifi", n_threads=0, **overrides):
        self.ctx, self.lib = ctx, load_library()
        _bind(self.lib)
        p = dict(PRESETS[preset])
        p.update(overrides)
        self.p = p
        n = len(seqs)
        # sequences: str / bytes (ASCII) or numpy uint8 arrays (ASCII or nt4 codes 0-3; seq_nt4_table maps both)
        keep = []
        ptrs = (C.c_void_p * n)()
        for i, s in enumerate(seqs):
            if isinstance(s, np.ndarray):
                s = np.ascontiguousarray(s, np.uint8)
                keep.append(s)
                ptrs[i] = s.ctypes.data
            else:
                s = s if isinstance(s, bytes) else s.encode()
                keep.append(s)
                ptrs[i] = C.cast(C.c_char_p(s), C.c_void_p).value
        names = [s if isinstance(s, bytes) else s.encode() for s in names]
        self.names = [x.decode() for x in names]
        a_names = (C.c_char_p * n)(*names)
        a_seqs = C.cast(ptrs, C.POINTER(C.c_char_p))
        lens = np.array([len(s) for s in keep], np.uint32)
        self._idx = C.c_void_p()
        rc = self.lib.gdiet_hip_index_build(ctx._h, C.byref(self._idx), n, a_names, a_seqs, lens.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            p["k"], p["w"], p["Z"].encode(), p["W"], n_threads)
        del keep
        ctx._check(rc)
        self.lens = lens
        self._setup_opt()

    @classmethod
    def from_flat(cls, ctx, names, lens, flat, preset="hifi", **overrides):
        """the gdiet_hip_index_import route: `flat` = dict(keys, cnt, pos, S, offsets) as a reference-side stub would
        produce by walking mm_idx_t::B[], or as export_index() returns"""
        self = cls.__new__(cls)
        self.ctx, self.lib = ctx, load_library()
        _bind(self.lib)
        p = dict(PRESETS[preset])
        p.update(overrides)
        self.p = p
        n = len(names)
        bn = [s if isinstance(s, bytes) else s.encode() for s in names]
        self.names = [x.decode() for x in bn]
        self.lens = np.ascontiguousarray(lens, np.uint32)
        u64, u32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        keys, cnt, pos = (np.ascontiguousarray(flat[k], t) for k, t in (("keys", np.uint64), ("cnt", np.uint32), ("pos", np.uint64)))
        S, offs = np.ascontiguousarray(flat["S"], np.uint32), np.ascontiguousarray(flat["offsets"], np.uint64)
        self._idx = C.c_void_p()
        rc = self.lib.gdiet_hip_index_import(ctx._h, C.byref(self._idx), p["k"], p["w"], p["Z"].encode(), p["W"], n, (C.c_char_p * n)(*bn),
                                             self.lens.ctypes.data_as(u32), offs.ctypes.data_as(u64), S.ctypes.data_as(u32), len(keys),
                                             keys.ctypes.data_as(u64), cnt.ctypes.data_as(u32), pos.ctypes.data_as(u64))
        ctx._check(rc)
        self._setup_opt()
        return self

    @classmethod
    def from_mmi(cls, ctx, path, names, lens, preset="hifi", **overrides):
        """gdiet_hip_index_load_mmi: an index file written by the reference (`GDiet -d`)"""
        self = cls.__new__(cls)
        self.ctx, self.lib = ctx, load_library()
        _bind(self.lib)
        p = dict(PRESETS[preset])
        p.update(overrides)
        self.p = p
        self.names, self.lens = list(names), np.ascontiguousarray(lens, np.uint32)
        self._idx = C.c_void_p()
        ctx._check(self.lib.gdiet_hip_index_load_mmi(ctx._h, C.byref(self._idx), path.encode(), p["Z"].encode(), p["W"]))
        self._setup_opt()
        return self

    def dump_mmi(self, path, bucket_bits=14):
        self.ctx._check(self.lib.gdiet_hip_index_dump_mmi(self.ctx._h, self._idx, path.encode(), bucket_bits))

    def export_index(self):
        u64, u32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        nk, npos, ns = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx._check(self.lib.gdiet_hip_index_export(self._idx, C.byref(nk), C.byref(npos), C.byref(ns), None, None, None, None, None))
        keys, cnt, pos = np.zeros(nk.value, np.uint64), np.zeros(nk.value, np.uint32), np.zeros(npos.value, np.uint64)
        S, offs = np.zeros(ns.value, np.uint32), np.zeros(len(self.names), np.uint64)
        self.ctx._check(self.lib.gdiet_hip_index_export(self._idx, None, None, None, keys.ctypes.data_as(u64), cnt.ctypes.data_as(u32),
                                                        pos.ctypes.data_as(u64), S.ctypes.data_as(u32), offs.ctypes.data_as(u64)))
        return dict(keys=keys, cnt=cnt, pos=pos, S=S, offsets=offs)

    def _setup_opt(self):
        p = self.p
        # mm_mapopt_update (reference options.c:64-76)
        mid = p.get("mid_occ", 0)
        if mid <= 0:
            mid = self.lib.gdiet_hip_index_cal_max_occ(self._idx, C.c_float(2e-4))
            mid = max(mid, p["min_mid_occ"])
            if p["max_mid_occ"] > p["min_mid_occ"]:
                mid = min(mid, p["max_mid_occ"])
        self.mid_occ = mid
        self.opt = MapOpt(flag=p["flag"], a=p["a"], b=p["b"], q=p["q"], e=p["e"], q2=p["q2"], e2=p["e2"], bw=p["bw"],
                          min_dp_max=p["min_dp_max"], best_n=p["best_n"], q_occ_frac=0.01, mid_occ=mid, max_max_occ=4095,
                          occ_dist=p["occ_dist"], max_frag_len=p.get("max_frag_len", 0), vt_dis=p["vt_dis"], vt_nb_loc=p["vt_nb_loc"], vt_cov=p["vt_cov"],
                          vt_f=p["vt_f"], vt_df1=p["vt_df1"], vt_df2=p["vt_df2"], max_max_gap=p["max_max_gap"],
                          max_min_gap=p["max_min_gap"], max_seeds=p["max_seeds"], min_cnt=p.get("min_cnt", 1.0),
                          rec_threshold_frac=p.get("rec_threshold_frac", 0.0), bw_frac=p.get("bw_frac", 0.05), bw_min=p.get("bw_min", 500),
                          bw_max=p.get("bw_max", 1500), AF_max_loc=p.get("AF_max_loc", 20))

    def close(self):
        if getattr(self, "_sam_buf", None) is not None and self._sam_buf.value:
            C.CDLL(None).free(self._sam_buf)
            self._sam_buf = C.c_void_p()
        if self._idx:
            self.lib.gdiet_hip_index_destroy(self.ctx._h, self._idx)
            self._idx = C.c_void_p()

    def n_keys(self):
        return self.lib.gdiet_hip_index_n_keys(self._idx)

    def _arrays(self, reads):
        n = len(reads)
        reads = [s if isinstance(s, bytes) else s.encode() for s in reads]
        return n, reads, (C.c_char_p * n)(*reads), np.array([len(s) for s in reads], np.int32)

    def map(self, reads):
        n, reads, arr, lens = self._arrays(reads)
        n_regs = (C.c_int32 * n)()
        regs = (C.POINTER(Reg) * n)()
        rc = self.lib.gdiet_hip_map_batch(self.ctx._h, self._idx, C.byref(self.opt), n, arr, lens.ctypes.data_as(C.POINTER(C.c_int32)), n_regs, regs)
        self.ctx._check(rc)
        return MapResult(self.lib, n, n_regs, regs)

    def upload(self, reads):
        n, reads, arr, lens = self._arrays(reads)
        h = C.c_void_p()
        self.ctx._check(self.lib.gdiet_hip_batch_upload(self.ctx._h, C.byref(h), n, arr, lens.ctypes.data_as(C.POINTER(C.c_int32))))
        return (h, n)

    def upload_raw(self, n, seqs, lens):
        """gdiet_hip_batch_upload on C arrays as they are (e.g. the ones gdiet_hip_fastx_read returned): no Python object per read"""
        h = C.c_void_p()
        self.ctx._check(self.lib.gdiet_hip_batch_upload(self.ctx._h, C.byref(h), n, C.cast(seqs, C.POINTER(C.c_char_p)), lens))
        return (h, n)

    def sam_batch_raw(self, res, n, names, seqs, quals, lens, sink=None):
        """gdiet_hip_sam_batch on C arrays.  With sink (a binary file object) the text is formatted into a buffer kept by this mapper
        (gdiet_hip_sam_batch_into) and written from it without a copy, and its length returned; otherwise bytes are returned."""
        cpp = C.POINTER(C.c_char_p)
        if sink is not None:
            if not hasattr(self, "_sam_buf"):
                self._sam_buf, self._sam_cap = C.c_void_p(), C.c_size_t(0)
                self.lib.gdiet_hip_sam_batch_into.restype = C.c_size_t
                self.lib.gdiet_hip_sam_batch_into.argtypes = self.lib.gdiet_hip_sam_batch.argtypes[:-1] + [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
            m = self.lib.gdiet_hip_sam_batch_into(self.ctx._h, self._idx, n, C.cast(names, cpp), C.cast(seqs, cpp), C.cast(quals, cpp), lens, res.n_regs, res.regs,
                                                  self.opt.flag, C.byref(self._sam_buf), C.byref(self._sam_cap))
            if m:
                sink.write(memoryview((C.c_char * m).from_address(self._sam_buf.value)))
            return m
        out = C.c_void_p()
        m = self.lib.gdiet_hip_sam_batch(self.ctx._h, self._idx, n, C.cast(names, cpp), C.cast(seqs, cpp), C.cast(quals, cpp), lens, res.n_regs, res.regs,
                                         self.opt.flag, C.byref(out))
        try:
            return C.string_at(out.value, m) if out.value else b""
        finally:
            if out.value:
                C.CDLL(None).free(C.c_void_p(out.value))

    def map_uploaded(self, batch):
        h, n = batch
        n_regs = (C.c_int32 * n)()
        regs = (C.POINTER(Reg) * n)()
        self.ctx._check(self.lib.gdiet_hip_map_uploaded(self.ctx._h, self._idx, C.byref(self.opt), h, n_regs, regs))
        return MapResult(self.lib, n, n_regs, regs)

    def set_inflight(self, n):
        self.ctx._check(self.lib.gdiet_hip_set_inflight(self.ctx._h, n))

    def submit(self, batch):
        """start mapping a resident batch (gdiet_hip_map_submit); returns a ticket for wait().  At most set_inflight() tickets (default 2, at most 8) may be open."""
        h, n = batch
        n_regs = (C.c_int32 * n)()
        regs = (C.POINTER(Reg) * n)()
        t = C.c_void_p()
        self.ctx._check(self.lib.gdiet_hip_map_submit(self.ctx._h, self._idx, C.byref(self.opt), h, n_regs, regs, C.byref(t)))
        return (t, n, n_regs, regs)

    def wait(self, ticket):
        t, n, n_regs, regs = ticket
        self.ctx._check(self.lib.gdiet_hip_map_wait(self.ctx._h, t))
        return MapResult(self.lib, n, n_regs, regs)

    def set_lanes(self, n):
        """software-pipeline depth of map()/map_uploaded() (gdiet_hip_set_map_lanes); results do not depend on it"""
        self.ctx._check(self.lib.gdiet_hip_set_map_lanes(self.ctx._h, n))

    def set_host_threads(self, n):
        self.ctx._check(self.lib.gdiet_hip_set_host_threads(self.ctx._h, n))

    def free_batch(self, batch):
        self.lib.gdiet_hip_batch_destroy(self.ctx._h, batch[0])

    def stage_seconds(self):
        out = (C.c_double * 6)()
        self.lib.gdiet_hip_map_stage_seconds(self.ctx._h, out)
        return list(out)

    def sam_batch(self, res, reads):
        """every SAM record of a MapResult as one string (gdiet_hip_sam_batch); reads = [(qname, seq, qual or None), ...]"""
        n = len(reads)
        enc = lambda x: x if isinstance(x, bytes) else x.encode()
        qn = (C.c_char_p * n)(*[enc(r[0]) for r in reads])
        sq = (C.c_char_p * n)(*[enc(r[1]) for r in reads])
        ql = (C.c_char_p * n)(*[None if len(r) < 3 or r[2] is None else enc(r[2]) for r in reads])
        lens = np.array([len(r[1]) for r in reads], np.int32)
        out = C.c_void_p()
        m = self.lib.gdiet_hip_sam_batch(self.ctx._h, self._idx, n, qn, sq, ql, lens.ctypes.data_as(C.POINTER(C.c_int32)), res.n_regs, res.regs,
                                         self.opt.flag, C.byref(out))
        try:
            return C.string_at(out.value, m).decode() if out.value else ""
        finally:
            if out.value:
                C.CDLL(None).free(C.c_void_p(out.value))

    def paf_batch(self, res, reads, flag=0):
        """every PAF line of a MapResult as one string (gdiet_hip_paf_batch); reads = [(qname, seq, ...), ...]; flag: extra MM_F_* bits
        (0x20 = cg:Z: tag, 0x8000000 = lines for unmapped reads)"""
        n = len(reads)
        enc = lambda x: x if isinstance(x, bytes) else x.encode()
        qn = (C.c_char_p * n)(*[enc(r[0]) for r in reads])
        lens = np.array([len(r[1]) for r in reads], np.int32)
        out = C.c_void_p()
        m = self.lib.gdiet_hip_paf_batch(self.ctx._h, self._idx, n, qn, lens.ctypes.data_as(C.POINTER(C.c_int32)), res.n_regs, res.regs,
                                         self.opt.flag | flag, C.byref(out))
        try:
            return C.string_at(out.value, m).decode() if out.value else ""
        finally:
            if out.value:
                C.CDLL(None).free(C.c_void_p(out.value))

    def sam(self, res, i, qname, seq, qual=None):
        """SAM lines of read i of a MapResult, as the reference's output step prints them (map.c step 2)."""
        lines = []
        seq_b = seq if isinstance(seq, bytes) else seq.encode()
        qual_b = None if qual is None else (qual if isinstance(qual, bytes) else qual.encode())
        n = res.n_regs[i]
        buf = C.create_string_buffer(4 * len(seq_b) + 4096 + 64 * max(1, sum(res.regs[i][j].n_cigar for j in range(n))))
        idxs = list(range(n)) if n > 0 else [-1]
        for j in idxs:
            if j >= 0 and (self.opt.flag & F_NO_PRINT_2ND) and res.regs[i][j].id != res.regs[i][j].parent:
                continue
            need = self.lib.gdiet_hip_sam_record(self._idx, qname.encode(), seq_b, qual_b, len(seq_b), res.regs[i], n, j, self.opt.flag, buf, len(buf))
            if need >= len(buf):
                raise GdietError("SAM buffer too small")
            lines.append(buf.value.decode())
        return lines
