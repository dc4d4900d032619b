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
    read -> upload + submit -> wait + format + write; the C calls release the interpreter lock, so the stages overlap.
    A failure in any stage stops all three: `stop` is set, the writer keeps draining q_done (every open ticket is still waited for and
    its permit, batch and reader arena released -- the lanes would otherwise keep running against the context), the reader and the
    submitting loop give up at their next queue operation, and the first error is raised once the threads have ended."""
    import queue
    import threading
    fx = pkg.FastxReader(reads_path, threads=reader_threads)
    mapper.set_inflight(inflight)
    T = {"read": 0.0, "upload": 0.0, "submit": 0.0, "wait": 0.0, "sam+write": 0.0, "free": 0.0}
    q_read, q_done = queue.Queue(maxsize=2), queue.Queue()
    open_tickets = threading.Semaphore(inflight)  # the library takes at most `inflight` tickets: one permit per open ticket
    errors, stop = [], threading.Event()
    t0 = time.perf_counter()

    def timed(key, f, *a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        T[key] += time.perf_counter() - t
        return r

    def fail(e):
        errors.append(e)
        stop.set()

    def put(q, item):
        """q.put that gives up once the pipeline is stopping; False = not queued"""
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                pass
        return False

    def reader():
        try:
            while not stop.is_set():
                item = timed("read", fx.read_raw, chunk, detach=True)
                if not put(q_read, item):
                    if item[0]:
                        fx.release(item[6])
                    return
                if item[0] == 0:
                    return
        except Exception as e:  # noqa: BLE001
            fail(e)

    def writer():
        while True:
            item = q_done.get()
            if item is None:
                return
            ticket, token, n, names, seqs, quals, lens, batch = item
            res = None
            try:  # the ticket is waited for whatever happened before: only then is its lane idle
                res = timed("wait", mapper.wait, ticket)
            except Exception as e:  # noqa: BLE001
                fail(e)
            open_tickets.release()
            try:
                if res is not None and not stop.is_set():
                    timed("sam+write", mapper.sam_batch_raw, res, n, names, seqs, quals, lens, out)
            except Exception as e:  # noqa: BLE001
                fail(e)
            try:
                def drop():
                    nonlocal res
                    res = None
                    mapper.free_batch(batch)
                    fx.release(token)
                timed("free", drop)
            except Exception as e:  # noqa: BLE001
                fail(e)

    th_r, th_w = threading.Thread(target=reader), threading.Thread(target=writer)
    th_r.start(), th_w.start()
    n_reads = 0
    try:
        while not stop.is_set():
            try:
                n, names, comments, seqs, quals, lens, token = q_read.get(timeout=0.2)
            except queue.Empty:
                if not th_r.is_alive() and q_read.empty():
                    break  # the reader ended without its end-of-file item: it failed
                continue
            if n == 0:
                break
            n_reads += n
            batch = None
            try:
                batch = timed("upload", mapper.upload_raw, n, seqs, lens)
                while not open_tickets.acquire(timeout=0.2):
                    if stop.is_set():
                        raise RuntimeError("pipeline stopped")
                try:
                    ticket = timed("submit", mapper.submit, batch)
                except Exception:
                    open_tickets.release()
                    raise
                q_done.put((ticket, token, n, names, seqs, quals, lens, batch))
            except Exception as e:  # noqa: BLE001
                if batch is not None:
                    mapper.free_batch(batch)
                fx.release(token)
                if not stop.is_set() or not errors:
                    fail(e)
                break
    finally:
        stop_was_set = stop.is_set()
        q_done.put(None)  # the writer drains what is queued before it, then ends
        if stop_was_set:  # unblock a reader waiting on a full queue
            while True:
                try:
                    item = q_read.get_nowait()
                    if item[0]:
                        fx.release(item[6])
                except queue.Empty:
                    break
        th_w.join()
        stop.set()  # (nothing is left to do: a reader that has not reached the end of the file stops here)
        th_r.join()
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
    if a.ref.startswith("synth:"):  # "synth:3088": bench.py's synthetic reference of that many Mbp, made in-process (no 3 GB FASTA file to write and parse)
        import bench
        names, seqs = bench.synth_reference(float(a.ref.split(":")[1]), seed=2)
    else:
        names, seqs = read_fasta(a.ref)
    t0 = time.perf_counter()
    m = pkg.Mapper(ctx, names, seqs, preset=a.preset, n_threads=pkg.effective_cpus())
    m.set_host_threads(pkg.effective_cpus())
    t_idx = time.perf_counter() - t0
    chunk = a.K or (39321600 if a.preset == "sr" else 80_000_000)
    try:
        with open(a.out, "wb") as out:
            n, dt = map_file(pkg, m, a.reads, out, chunk, a.inflight, a.reader_threads)
    except Exception as e:  # noqa: BLE001  (every ticket has been waited for by now: the context can be closed)
        print("map_file failed: %r" % (e,), file=sys.stderr)
        m.close()
        ctx.close()
        sys.exit(1)
    print(json.dumps({"reads": n, "seconds": round(dt, 3), "reads_per_s": round(n / dt), "index_s": round(t_idx, 2), "mini_batch_bases": chunk,
                      "inflight": a.inflight, "reader_threads": a.reader_threads, "out": a.out, "caller_seconds": map_file.last_stage_seconds}))
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
