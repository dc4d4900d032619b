"""Steps 0-2 of the reference's worker_pipeline through the library: FASTA/FASTQ file -> mini-batches (gdiet_hip_fastx_read) ->
map with several batches in flight (gdiet_hip_map_submit / _wait) -> SAM records (gdiet_hip_sam_batch) -> output file.
No Python object is made per read: the C arrays of the reader go straight into the upload and the SAM formatter.

    python tools/map_file.py --preset sr ref.fa reads.fq[.gz] -o out.sam [-K 39321600] [--inflight 3] [--reader-threads 4]

Writes the SAM body (the records; the header lines of the reference's CLI are not part of the path)."""
import argparse
import json
import os
import sys
import time

import torch  # noqa: F401  (first: one HIP runtime for torch and libgdiet_hip.so)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_pkg  # noqa: E402


def map_file(pkg, mapper, reads_path, out, chunk, inflight, reader_threads):
    """returns (reads, seconds).  Three stages on three threads, as the reference's kt_pipeline runs its three steps (LR/map.c:2094-2170):
    read -> upload + submit -> wait + format + write; the C calls release the interpreter lock, so the stages overlap."""
    import queue
    import threading
    fx = pkg.FastxReader(reads_path, threads=reader_threads)
    mapper.set_inflight(inflight)
    T = {"read": 0.0, "upload": 0.0, "submit": 0.0, "wait": 0.0, "sam+write": 0.0, "free": 0.0}
    q_read, q_done = queue.Queue(maxsize=2), queue.Queue()
    open_tickets = threading.Semaphore(inflight)  # the library takes at most `inflight` tickets: one permit per open ticket
    errors = []
    t0 = time.perf_counter()

    def timed(key, f, *a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        T[key] += time.perf_counter() - t
        return r

    def reader():
        try:
            while True:
                item = timed("read", fx.read_raw, chunk, detach=True)
                q_read.put(item)
                if item[0] == 0:
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            q_read.put((0,) + (None,) * 6)

    def writer():
        try:
            while True:
                item = q_done.get()
                if item is None:
                    return
                ticket, token, n, names, seqs, quals, lens, batch = item
                res = timed("wait", mapper.wait, ticket)
                open_tickets.release()
                timed("sam+write", mapper.sam_batch_raw, res, n, names, seqs, quals, lens, out)

                def drop():
                    nonlocal res
                    res = None
                    mapper.free_batch(batch)
                    fx.release(token)
                timed("free", drop)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th_r, th_w = threading.Thread(target=reader), threading.Thread(target=writer)
    th_r.start(), th_w.start()
    n_reads = 0
    while True:
        n, names, comments, seqs, quals, lens, token = q_read.get()
        if n == 0:
            break
        n_reads += n
        batch = timed("upload", mapper.upload_raw, n, seqs, lens)
        open_tickets.acquire()
        q_done.put((timed("submit", mapper.submit, batch), token, n, names, seqs, quals, lens, batch))
    q_done.put(None)
    th_r.join(), th_w.join()
    fx.close()
    if errors:
        raise errors[0]
    map_file.last_stage_seconds = {k: round(v, 3) for k, v in T.items()}
    return n_reads, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ref")
    ap.add_argument("reads")
    ap.add_argument("-o", "--out", default="/dev/null")
    ap.add_argument("--preset", default="hifi", choices=["hifi", "ont", "sr"])
    ap.add_argument("-K", type=int, default=0, help="bases per mini-batch (default: 39321600 for sr = 262144 reads of 150, 80e6 for long reads)")
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--reader-threads", type=int, default=4)
    a = ap.parse_args()
    pkg = _load_pkg()
    from fixture_io import read_fasta
    ctx = pkg.Context(0)
    names, seqs = read_fasta(a.ref)
    t0 = time.perf_counter()
    m = pkg.Mapper(ctx, names, seqs, preset=a.preset, n_threads=pkg.effective_cpus())
    m.set_host_threads(pkg.effective_cpus())
    t_idx = time.perf_counter() - t0
    chunk = a.K or (39321600 if a.preset == "sr" else 80_000_000)
    with open(a.out, "wb") as out:
        n, dt = map_file(pkg, m, a.reads, out, chunk, a.inflight, a.reader_threads)
    print(json.dumps({"reads": n, "seconds": round(dt, 3), "reads_per_s": round(n / dt), "index_s": round(t_idx, 2), "mini_batch_bases": chunk,
                      "inflight": a.inflight, "reader_threads": a.reader_threads, "out": a.out, "caller_seconds": map_file.last_stage_seconds}))
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
