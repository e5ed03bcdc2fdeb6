"""Stitch behind a MULTI-RANK inference: collector processes that take the ranks' regions as they are decoded.

With one rank, `helen polish` decodes regions and aligns neighbours in the rank itself while its device works
(helen_amd.stitch_stream).  With one rank per GPU that is not enough.  The reference deals the image FILES to the ranks
(CallConsensusInterface.py:138-145) and MarginPolish writes a contig's regions into whichever thread's file, so the two
neighbours of most joins are decoded by different ranks; and eight ranks finish eight times the regions in the same
time.  Left to the end of the run, the parent would load every rank's sequences (gigabytes), align most joins and walk
every region in one interpreter: on a whole genome that tail is twice the device time.

So the work is sharded by CONTIG instead.  The parent starts W collector processes before the ranks.  A rank only
decodes: its RegionStream writes each region -- (contig, start, end, sequence) -- as one record into the append-only file
of the collector that owns the contig (crc32 of the name mod W; files under /dev/shm, one per (rank, collector)).  A
collector follows its R files while the ranks run, keeps the regions per rank (the order of the prediction FILES decides
between regions of one span, as in perform_stitch), and runs the same neighbour speculation as a single rank's stream, on
its share of the `-t` threads -- over the regions of ALL ranks, so neighbours from different ranks meet here.  When every
rank has written its end marker the collector joins its contigs (helen_amd.stitch_stream.assemble_contigs: slices where
every join is ordinary, the reference's own order otherwise) into a part file with an index, and the parent copies the
parts into the FASTA in sorted contig order.  What is serial after the last window is that copy.

The FASTA is perform_stitch's (tests/test_stitch_collect.py: the same adversarial streams as the single-rank tests, dealt
over several ranks and collectors; tests/test_polish_chain.py runs the command with two callers).
"""
import json
import multiprocessing as mp
import os
import struct
import sys
import time
import zlib

_HEADER = struct.Struct("<IqqI")
_NO_SEQUENCE = 0xFFFFFFFF          # seq_len of a record that says "decode this region from the prediction file"
_END = 0xFFFFFFFF                  # contig_len of the record that ends a rank's file


def bucket_of(contig, buckets):
    return zlib.crc32(contig.encode()) % buckets


def _path(prefix, rank, bucket):
    return "%s_%d_%d.bin" % (prefix, rank, bucket)


class RegionExport(object):
    """The rank's side: records appended to one file per collector.  The files exist before any rank starts (the parent
    creates them), so a collector never waits for a name to appear."""

    def __init__(self, prefix, rank, buckets):
        self.buckets = buckets
        self.fds = [os.open(_path(prefix, rank, w), os.O_WRONLY | os.O_APPEND) for w in range(buckets)]

    def _send(self, per_bucket):
        for w, parts in per_bucket.items():
            data = b"".join(parts)
            view = memoryview(data)
            while len(view):
                view = view[os.write(self.fds[w], view):]

    def write(self, keys, seqs):
        per_bucket = {}
        for (contig, start, end), seq in zip(keys, seqs):
            name = contig.encode()
            parts = per_bucket.setdefault(bucket_of(contig, self.buckets), [])
            parts.append(_HEADER.pack(len(name), start, end, len(seq)))
            parts.append(name)
            parts.append(seq)
        self._send(per_bucket)

    def from_file(self, key):
        name = key[0].encode()
        self._send({bucket_of(key[0], self.buckets): [_HEADER.pack(len(name), key[1], key[2], _NO_SEQUENCE), name]})

    def close(self):
        """The end marker: this rank has decoded its last region (and closed its prediction file)."""
        for fd in self.fds:
            os.write(fd, _HEADER.pack(_END, 0, 0, 0))
            os.close(fd)
        self.fds = []

    def abandon(self):
        """A failed rank: no end marker (the parent takes the collectors down)."""
        for fd in self.fds:
            os.close(fd)
        self.fds = []


_FALLOC_FL_KEEP_SIZE, _FALLOC_FL_PUNCH_HOLE = 1, 2
_libc = [None]


def _punch_hole(fd, length):
    """Give the pages of bytes [0, length) of a record file back (tmpfs pages are RAM): the records are in the collector's
    heap by now.  Best effort -- a file system without hole punching keeps them until the run's directory goes."""
    try:
        if _libc[0] is None:
            import ctypes
            lib = ctypes.CDLL(None, use_errno=True)
            lib.fallocate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong]
            lib.fallocate.restype = ctypes.c_int
            _libc[0] = lib
        return _libc[0].fallocate(fd, _FALLOC_FL_PUNCH_HOLE | _FALLOC_FL_KEEP_SIZE, 0, length) == 0
    except (OSError, AttributeError):
        _libc[0] = False
        return False


class _Follower(object):
    """One rank's file as the collector reads it: whatever has been appended since the last look, cut into records."""

    def __init__(self, path):
        self.fd = os.open(path, os.O_RDWR)
        self.rest = b""
        self.ended = False
        self.read_bytes = 0
        self.punched = 0

    def poll(self):
        """-> ([(key, sequence or None)], bytes read)."""
        chunks = []
        got = 0
        while True:
            data = os.read(self.fd, 1 << 24)
            if not data:
                break
            chunks.append(data)
            got += len(data)
            if got >= 1 << 26:
                break
        if not chunks:
            return [], 0
        self.read_bytes += got
        if _libc[0] is not False and self.read_bytes - self.punched >= 1 << 26:     # every 64 MiB, whole pages
            upto = self.read_bytes & ~4095
            if _punch_hole(self.fd, upto):
                self.punched = upto
        buf = self.rest + b"".join(chunks)
        out, at, n = [], 0, len(buf)
        size = _HEADER.size
        while at + size <= n:
            name_len, start, end, seq_len = _HEADER.unpack_from(buf, at)
            if name_len == _END:
                self.ended = True
                at = n
                break
            body = name_len + (0 if seq_len == _NO_SEQUENCE else seq_len)
            if at + size + body > n:
                break
            name = buf[at + size:at + size + name_len].decode()
            seq = None if seq_len == _NO_SEQUENCE else buf[at + size + name_len:at + size + body]
            out.append(((name, start, end), seq))
            at += size + body
        self.rest = buf[at:]
        return out, got

    def close(self):
        os.close(self.fd)


def _collector_main(bucket, buckets, prefix, prediction_files, threads, run_threads, part_path, result_q):
    """A collector process: follow the ranks' files of this bucket, speculate joins, assemble the bucket's contigs."""
    import signal

    def on_term(signum, frame):
        raise SystemExit(143)
    signal.signal(signal.SIGTERM, on_term)
    import gc
    gc.disable()        # millions of tuples and byte strings, no cycles: the collector's generations only cost time here
    from . import stitch_stream
    ranks = len(prediction_files)
    followers = [_Follower(_path(prefix, r, bucket)) for r in range(ranks)]
    # ONE stream over the regions of all ranks (neighbours by position, joins on this collector's threads) ...
    stream = stitch_stream.RegionStream(prediction_files[0], threads)
    # ... and the regions per rank, for the order between regions of one span and for the regions to read back
    per_rank = [{} for _ in range(ranks)]
    t0 = time.time()
    busy = 0.0
    from_file = 0
    parent = os.getppid()
    while not all(f.ended for f in followers):
        if os.getppid() != parent:              # the run's parent is gone (killed): nobody will ever write the end markers
            raise SystemExit(1)
        progressed = False
        for r, f in enumerate(followers):
            if f.ended:
                continue
            records, got = f.poll()
            if not got:
                continue
            progressed = True
            t1 = time.time()
            keys, seqs = [], []
            for key, seq in records:
                if seq is None:
                    per_rank[r][key] = None
                    if stream.regions.get(key, b"") is not None:
                        stream.regions[key] = None
                    from_file += 1
                    continue
                per_rank[r][key] = seq
                if key in stream.regions:
                    continue                    # a second rank holds a region of this span: the join pass sorts it out
                keys.append(key)
                seqs.append(seq)
            if keys:
                stream.accept_sequences(keys, seqs)
            busy += time.time() - t1
        if not progressed:
            time.sleep(0.01)
    t_last = time.time()
    for f in followers:
        f.close()
    result = stream.finish()
    t_joined = time.time()
    index = []
    writer = stitch_stream.FastaWriter(part_path)
    offset = [0]

    def emit(contig, pieces):
        n = sum(len(p) for p in pieces)
        index.append((contig, offset[0], n))
        offset[0] += n
        writer.put(pieces)
    try:
        by_file = {os.path.abspath(p): per_rank[r] for r, p in enumerate(prediction_files)}
        # between regions of one span the order of the FILES decides, and that is the directory's listing order
        # (StitchInterface.py:35-36 takes os.listdir as it comes) -- not the order of the ranks
        listing = [p for p in stitch_stream.get_file_paths_from_directory(os.path.dirname(prediction_files[0]))]
        if sorted(os.path.abspath(p) for p in listing) != sorted(by_file):
            raise RuntimeError("the prediction directory holds other files than this run's (%s): run `helen stitch` on it"
                               % sorted(set(os.path.abspath(p) for p in listing) ^ set(by_file))[:2])
        stats = stitch_stream.assemble_contigs(listing, by_file, result.joins, result.pair_joins, threads, emit,
                                               run_threads=run_threads, quiet=True)
        t_assembled = time.time()
    finally:
        writer.close()
    from .host_plan import peak_rss_mb, rss_breakdown_mb
    stats.update({"bucket": bucket, "regions": sum(len(d) for d in per_rank), "from_file": from_file, "peak_rss_mb": peak_rss_mb(),
                  "rss_anon_mb": rss_breakdown_mb()[0],
                  "joins_submitted": result.stats.get("joins_submitted", 0), "index": index,
                  "seconds": {"following": round(t_last - t0, 3), "accepting": round(busy, 3),
                              "joins_after_the_last_record": round(t_joined - t_last, 3),
                              "assembly": round(time.time() - t_joined, 3),
                              "of_which_waiting_for_the_part_file": round(time.time() - t_assembled, 3)}})
    result_q.put((bucket, stats))


def collectors_for(threads):
    """How many collector processes a run with `-t threads` starts: one per four threads, eight at most."""
    return max(1, min(8, int(threads) // 4))


class CollectorRun(object):
    """The parent's handle: start() before the ranks, finish() after them (-> the FASTA), abort() when a rank failed."""

    def __init__(self, prediction_files, threads, directory=None, expected_bytes=0):
        """`expected_bytes`: what the run will park between ranks and collectors -- the called sequences once as records
        (given back page by page as the collectors read them) and once as part files, about two bytes per window position.
        The files go to a directory of this run's own (removed with the run) under /dev/shm when that may take them
        (helen_amd.stitch_stream.spill_directory: free space and half of the available RAM, at least 1 GiB -- Docker's
        default 64 MB /dev/shm is never used), under the prediction directory otherwise."""
        import tempfile

        from .stitch_stream import spill_directory
        self.files = [os.path.abspath(p) for p in prediction_files]
        self.threads = max(1, int(threads))
        self.buckets = collectors_for(self.threads)
        parent = directory or spill_directory(expected_bytes) or os.path.dirname(self.files[0])
        self.sweep(parent)
        self.directory = tempfile.mkdtemp(prefix="helen_regions_", dir=parent)
        self.lock = open(os.path.join(self.directory, "lock"), "w")
        try:
            import fcntl
            fcntl.flock(self.lock, fcntl.LOCK_EX | fcntl.LOCK_NB)       # held while the run lives: sweep() leaves it alone
        except (ImportError, OSError):
            pass
        self.prefix = os.path.join(self.directory, "r")
        self.parts = [self.prefix + "_part%d.fa" % w for w in range(self.buckets)]
        self.procs = []
        self.result_q = None
        self.stats = None

    def export_spec(self):
        """What a rank needs to write its regions: (prefix, buckets)."""
        return (self.prefix, self.buckets)

    @staticmethod
    def sweep(parent, min_age_seconds=600.0):
        """Directories a KILLED run left behind under `parent`: provably stale only -- nobody holds the run's lock (a live
        run does, whatever PID namespace it is in: containers sharing /dev/shm) and nothing in it changed for ten minutes."""
        import shutil
        try:
            names = os.listdir(parent)
        except OSError:
            return 0
        removed = 0
        for name in names:
            d = os.path.join(parent, name)
            if not name.startswith("helen_regions_") or not os.path.isdir(d):
                continue
            try:
                newest = max([os.stat(d).st_mtime] + [os.stat(os.path.join(d, n)).st_mtime for n in os.listdir(d)])
                if time.time() - newest < min_age_seconds:
                    continue
                with open(os.path.join(d, "lock"), "a") as lock:
                    import fcntl
                    fcntl.flock(lock, fcntl.LOCK_EX | fcntl.LOCK_NB)    # raises while the run lives
                    shutil.rmtree(d, ignore_errors=True)
                    removed += 1
            except (OSError, ImportError):
                continue
        return removed

    def start(self):
        for r in range(len(self.files)):
            for w in range(self.buckets):
                open(_path(self.prefix, r, w), "wb").close()
        ctx = mp.get_context("spawn")
        self.result_q = ctx.Queue()
        share = max(1, self.threads // self.buckets)
        for w in range(self.buckets):
            p = ctx.Process(target=_collector_main, args=(w, self.buckets, self.prefix, self.files, share, self.threads,
                                                          self.parts[w], self.result_q), daemon=True)
            p.start()
            self.procs.append(p)
        return self

    def _cleanup(self):
        for r in range(len(self.files)):
            for w in range(self.buckets):
                try:
                    os.unlink(_path(self.prefix, r, w))
                except OSError:
                    pass
        for p in self.parts:
            try:
                os.unlink(p)
            except OSError:
                pass
        import shutil
        try:
            self.lock.close()
        except OSError:
            pass
        shutil.rmtree(self.directory, ignore_errors=True)

    def abort(self):
        for p in self.procs:
            if p.is_alive():
                p.terminate()
        for p in self.procs:
            p.join(5.0)
            if p.is_alive():
                p.kill()
                p.join()
        self.procs = []
        self._cleanup()

    def finish(self, output_path, output_prefix):
        """Wait for the collectors (the ranks have written their end markers) and copy their parts into
        `<output_path>/<output_prefix>.fa` in sorted contig order.  -> the FASTA's path."""
        import queue

        from . import file_manager
        from .stitch_stream import report_line
        t0 = time.time()
        got = {}
        try:
            while len(got) < self.buckets:
                try:
                    w, stats = self.result_q.get(timeout=0.2)
                    got[w] = stats
                except queue.Empty:
                    dead = [p for p in self.procs if not p.is_alive() and p.exitcode not in (0, None)]
                    if dead:
                        raise RuntimeError("a stitch collector exited with %s: the prediction files are complete, run "
                                           "`helen stitch` on them" % dead[0].exitcode)
                    if time.time() - t0 > 3600.0:       # (every rank has ended by now: an hour is not a backlog)
                        raise RuntimeError("the stitch collectors did not finish: the prediction files are complete, run "
                                           "`helen stitch` on them")
            for p in self.procs:
                p.join()
            t_collected = time.time()
            output_dir = file_manager.handle_output_directory(output_path)
            output_filename = os.path.join(output_dir, output_prefix + '.fa')
            sys.stderr.write("INFO: OUTPUT FILE: " + output_filename + "\n")
            where = {}
            for w, stats in got.items():
                for contig, offset, n in stats["index"]:
                    where[contig] = (w, offset, n)
            sources = {}
            with open(output_filename, "wb") as out:
                for contig in sorted(where):
                    w, offset, n = where[contig]
                    if w not in sources:
                        sources[w] = open(self.parts[w], "rb")
                    _copy_range(sources[w], out, offset, n)
            for f in sources.values():
                f.close()
        finally:
            self.abort()
        total = {"from_table": 0, "aligned_now": 0, "sliced": 0, "contigs": 0, "late": 0, "shortcut": 0, "three_passes": 0}
        for stats in got.values():
            for k in total:
                total[k] += stats[k]
        sys.stderr.write(report_line(total, sum(s["from_file"] for s in got.values()), self.threads))
        sys.stderr.write("INFO: %d STITCH COLLECTOR(S) OVER %d RANK(S): %d REGION(S); AFTER THE LAST REGION: JOINS %.2f S, "
                         "ASSEMBLY %.2f S (THE SLOWEST), FASTA COPY %.2f S.\n"
                         % (self.buckets, len(self.files), sum(s["regions"] for s in got.values()),
                            max(s["seconds"]["joins_after_the_last_record"] for s in got.values()),
                            max(s["seconds"]["assembly"] for s in got.values()), time.time() - t_collected))
        self.stats = {"collectors": self.buckets, "directory": os.path.dirname(self.directory),
                      "per_collector": [dict(got[w], index=len(got[w]["index"])) for w in sorted(got)],
                      "wait_seconds": round(t_collected - t0, 3), "copy_seconds": round(time.time() - t_collected, 3)}
        return output_filename

    def describe(self):
        return json.dumps(self.stats)


def _copy_range(src, dst, offset, n):
    """n bytes of `src` from `offset` on to the end of `dst` (both ordinary files): in the kernel where it can."""
    dst.flush()
    done = 0
    try:
        while done < n:
            k = os.copy_file_range(src.fileno(), dst.fileno(), n - done, offset + done)
            if k <= 0:
                break
            done += k
    except (AttributeError, OSError):
        pass
    if done < n:
        dst.seek(0, os.SEEK_END)
        src.seek(offset + done)
        while done < n:
            data = src.read(min(1 << 24, n - done))
            if not data:
                raise RuntimeError("a stitch collector's part file is shorter than its index says")
            dst.write(data)
            done += len(data)
    else:
        dst.seek(0, os.SEEK_END)
