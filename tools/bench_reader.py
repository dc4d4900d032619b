"""throughput of the FASTQ reader (gdiet_hip_fastx_*), CPU only: a synthetic 150 bp four-line FASTQ file, plain and gzip,
1..N parser threads.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    pkg = _load_pkg()
    lib = pkg.load_library()
    rng = np.random.default_rng(1)
    base = 500_000
    seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(base, 150))
    quals = rng.integers(35, 74, size=(base, 150), dtype=np.uint8)
    blob = b"".join(b"@read%d\n" % i + seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(base))
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    path = os.path.join(d, "big.fq")
    with open(path, "wb") as f:
        for _ in range(max(1, n_reads // base)):
            f.write(blob)
    size = os.path.getsize(path)
    cpp = C.POINTER(C.c_char_p)

    def run(p, threads):
        r = pkg.FastxReader(p, threads=threads)
        t0, tot = time.perf_counter(), 0
        while True:
            nn = C.c_int32()
            a, b, c, dd, e = cpp(), cpp(), cpp(), cpp(), C.POINTER(C.c_int32)()
            lib.gdiet_hip_fastx_read(r._h, 39_321_600, 1, 0, 0, C.byref(nn), C.byref(a), C.byref(b), C.byref(c), C.byref(dd), C.byref(e))
            if nn.value == 0:
                break
            tot += nn.value
        r.close()
        return tot, time.perf_counter() - t0

    out = {"file_MB": round(size / 1e6), "reads": None, "plain": {}, "cpus": pkg.effective_cpus()}
    for th in (1, 2, 4, 8):
        best = None
        for _ in range(3):
            tot, dt = run(path, th)
            best = dt if best is None else min(best, dt)
        out["reads"] = tot
        out["plain"]["threads_%d" % th] = {"MB_per_s": round(size / 1e6 / best), "M_reads_per_s": round(tot / best / 1e6, 2)}
    os.system("gzip -1 -k -f %s" % path)
    tot, dt = run(path + ".gz", 2)
    out["gzip"] = {"MB_per_s_uncompressed": round(size / 1e6 / dt), "M_reads_per_s": round(tot / dt / 1e6, 2)}
    os.remove(path), os.remove(path + ".gz"), os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
