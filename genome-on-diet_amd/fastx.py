"""ctypes mirror of the read-input entry points of include/gdiet_hip.h (gdiet_hip_fastx_*): FASTA / FASTQ, plain or gzip,
mini-batch by mini-batch with the reference's record grammar and batching rule (LR/bseq.c:80-121, LR/kseq.h:191-232)."""
import ctypes as C

from .hip_abi import GdietError, load_library

W_TRUNCATED = 1


class FastxReader:
    def __init__(self, path, threads=1):
        self.lib = load_library()
        L = self.lib
        cpp = C.POINTER(C.c_char_p)
        L.gdiet_hip_fastx_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        L.gdiet_hip_fastx_read.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(cpp), C.POINTER(cpp),
                                           C.POINTER(cpp), C.POINTER(cpp), C.POINTER(C.POINTER(C.c_int32))]
        L.gdiet_hip_fastx_close.argtypes = [C.c_void_p]
        L.gdiet_hip_fastx_close.restype = None
        self._h = C.c_void_p()
        if L.gdiet_hip_fastx_open(C.byref(self._h), path.encode() if isinstance(path, str) else path) != 0:
            raise GdietError("cannot open %r" % (path,))
        self.truncated = self.truncated_now = False
        L.gdiet_hip_fastx_set_threads.argtypes = [C.c_void_p, C.c_int]
        if threads > 1:
            L.gdiet_hip_fastx_set_threads(self._h, threads)

    def read(self, chunk_size, with_qual=True, with_comment=False, frag_mode=False):
        """next mini-batch as a list of (name, seq, qual or None, comment or None), all bytes; [] at the end of the input"""
        cpp = C.POINTER(C.c_char_p)
        n = C.c_int32()
        names, comments, seqs, quals, lens = cpp(), cpp(), cpp(), cpp(), C.POINTER(C.c_int32)()
        rc = self.lib.gdiet_hip_fastx_read(self._h, chunk_size, int(with_qual), int(with_comment), int(frag_mode), C.byref(n), C.byref(names),
                                           C.byref(comments), C.byref(seqs), C.byref(quals), C.byref(lens))
        if rc < 0:
            raise GdietError("read error")
        self.truncated_now = rc == W_TRUNCATED  # this batch was closed by a malformed record
        self.truncated = self.truncated or self.truncated_now
        out = []
        for i in range(n.value):
            s = seqs[i]
            assert len(s) == lens[i]
            out.append((names[i], s, quals[i], comments[i]))
        return out

    def read_raw(self, chunk_size, with_qual=True, with_comment=False, frag_mode=False, detach=False):
        """next mini-batch as C arrays: (n, names, comments, seqs, quals, lens, token).  With detach=True the arrays stay valid until
        release(token) (several mini-batches in flight); otherwise until the next read."""
        cpp = C.POINTER(C.c_char_p)
        n = C.c_int32()
        names, comments, seqs, quals, lens = cpp(), cpp(), cpp(), cpp(), C.POINTER(C.c_int32)()
        rc = self.lib.gdiet_hip_fastx_read(self._h, chunk_size, int(with_qual), int(with_comment), int(frag_mode), C.byref(n), C.byref(names),
                                           C.byref(comments), C.byref(seqs), C.byref(quals), C.byref(lens))
        if rc < 0:
            raise GdietError("read error")
        self.truncated_now = rc == W_TRUNCATED
        self.truncated = self.truncated or self.truncated_now
        token = None
        if detach and n.value:
            self.lib.gdiet_hip_fastx_detach.restype = C.c_void_p
            self.lib.gdiet_hip_fastx_detach.argtypes = [C.c_void_p]
            token = C.c_void_p(self.lib.gdiet_hip_fastx_detach(self._h))
        return n.value, names, comments, seqs, quals, lens, token

    def release(self, token):
        if token:
            self.lib.gdiet_hip_fastx_batch_free.argtypes = [C.c_void_p]
            self.lib.gdiet_hip_fastx_batch_free.restype = None
            self.lib.gdiet_hip_fastx_batch_free(token)

    def close(self):
        if self._h:
            self.lib.gdiet_hip_fastx_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
