"""Stitch pipelined behind inference: `helen polish` decodes regions and aligns neighbours WHILE the device stage runs.

The reference's `polish` is two phases -- call_consensus writes the prediction files, then perform_stitch reads them back
(PolishInterface.py:75-105) -- and with the inference on an MI355X the second phase is as long as the first.  Nothing in
stitch needs the whole prediction file before it starts:
  * a region's sequence (`small_chunk_stitch`'s position dictionaries, Stitch.py:204-247) depends on that region's own
    images only: RegionStream decodes it from the label buffers right after the writer stage has stored them (the
    prediction file is still written, byte for byte the same: stitch reads the buffers, not the file);
  * the overlap alignment of two neighbouring regions (Stitch.py:104-134) is a pure function of the two strings it is
    given: the tail of the running sequence and the head of the next region.  In every ordinary join the running tail IS
    the previous region's tail, so the alignment is computed speculatively by worker threads as soon as both regions are
    decoded, and kept in a table keyed by the two strings.
What is left for after the last window is `finish_stitch`: the reference's own join order -- regions sorted by
(start, end), runs of max(2, regions // threads + 1) regions stitched left to right, the runs stitched to each other
(Stitch.py:257-301) -- executed once, single-threaded, taking each alignment from the table when the strings it is
about to align are the speculated ones and from the aligner otherwise.  The FASTA is therefore the one perform_stitch
writes from the files (tests/test_stitch_stream.py compares the two on adversarial inputs), whatever was speculated.

A region whose images do not arrive back to back (possible only in a hand-made image file: MarginPolish writes a
region's images consecutively) is decoded from the finished prediction file instead, like perform_stitch does.
"""
import bisect
import operator
import concurrent.futures
import os
import pickle
import sys
import tempfile

import numpy as np

from . import file_manager, native_io
from .stitch import StitchOptions, _alignment_stitch, get_file_paths_from_directory

# images of one region a stream holds back at most (a region of more images is finished from the file)
_CARRY_LIMIT = 4096


def enabled():
    """Pipelining is the default of `polish`; $HELEN_STITCH_PIPELINE=0 runs the two phases one after the other."""
    return os.environ.get("HELEN_STITCH_PIPELINE", "1") != "0" and native_io.available()


_INJECT_FAILURE = [None]


class RegionStream(object):
    """The regions of ONE prediction file (one rank), decoded as the writer stage delivers their images.

    feed() is called from one thread, in the order the images are written; finish() after the last one.  `threads`
    worker threads run the speculative overlap alignments (native, the interpreter lock released)."""

    def __init__(self, prediction_file, threads=1, decode_threads=None, export=None):
        # export = a helen_amd.stitch_collect.RegionExport: this stream only DECODES, every region goes to the collector
        # process that owns its contig (a multi-rank run), which aligns and joins
        self.export = export
        self.file = os.path.abspath(prediction_file)
        self.threads = max(1, int(threads))
        self.decode_threads = max(1, min(8, self.threads)) if decode_threads is None else decode_threads
        self.regions = {}          # (contig, start, end) -> sequence bytes, or None = take it from the file
        self.written = {}          # (contig, start, end) -> set of chunk ids the writer has stored for it
        self.carry = None          # the newest region, still growing: [key, [(chunk id, positions, bases, rles), ...]]
        self.placed = {}           # contig -> (start, end) of its decoded regions, sorted
        self.joins = {}            # (left bytes, right bytes) -> (score, pos_a, pos_b)
        self.pair_joins = {}       # (contig, start a, end a, start b, end b) -> (score, pos_a, pos_b): the same results by REGION
        self.pool = concurrent.futures.ThreadPoolExecutor(self.threads) if native_io.available() and export is None else None
        self.pending = []
        self.seconds = {"decode": 0.0, "joins_submitted": 0, "regions": 0, "from_file": 0}

    # ---- feeding ----
    def feed(self, contigs, meta, positions, bases, rles):
        """`n` images in writing order: contigs uint8 [n, 256], meta int64 [n, 3] = contig_start, contig_end, chunk id,
        positions int64 [n, 1000, 3], bases / rles uint8 [n, 1000] -- the arrays DataStore.write_batch has just stored.
        Nothing of them is referenced after the call returns."""
        import time
        t0 = time.time()
        n = int(meta.shape[0])
        if n == 0:
            return
        if _INJECT_FAILURE[0] is None:      # test hook: the second feed of a stream fails as a full spill directory would
            _INJECT_FAILURE[0] = os.environ.get("HELEN_DEBUG_HOOKS") == "1" and os.environ.get("HELEN_DEBUG_STITCH_FAIL") == "1"
        if _INJECT_FAILURE[0]:
            self._feeds = getattr(self, "_feeds", 0) + 1
            if self._feeds == 2:
                raise OSError(28, "No space left on device (injected by HELEN_DEBUG_STITCH_FAIL)")
        contigs = np.asarray(contigs)
        # (a name ends at its first NUL; what follows in the slot's row may be left over from a longer name: only the
        # columns up to the longest name of this call are looked at, with every row's tail behind its NUL zeroed)
        is_nul = contigs == 0
        length = np.where(is_nul.any(axis=1), is_nul.argmax(axis=1), contigs.shape[1])
        width = int(length.max()) + 1 if n else 1
        names = contigs[:, :width] * (np.arange(width)[None, :] < length[:, None])
        # runs of consecutive images of one region
        change = np.ones(n, bool)
        if n > 1:
            change[1:] = (meta[1:, 0] != meta[:-1, 0]) | (meta[1:, 1] != meta[:-1, 1]) | \
                np.any(names[1:] != names[:-1], axis=1)
        seg_first = np.flatnonzero(change)
        seg_end = np.append(seg_first[1:], n)
        keys = []
        starts, ends = meta[seg_first, 0].tolist(), meta[seg_first, 1].tolist()
        last_row, last_name = None, None
        for a, cs, ce in zip(seg_first.tolist(), starts, ends):
            row = names[a].tobytes()
            if row != last_row:                       # (consecutive regions nearly always share their contig)
                last_row, last_name = row, row[:int(length[a])].decode()
            keys.append((last_name, cs, ce))
        # a region that shows up in two separate runs of this call (its images are not back to back): from the file
        count = {}
        for k in keys:
            count[k] = count.get(k, 0) + 1
        for k, c in count.items():
            if c > 1 and self.regions.get(k, b"") is not None:
                self._from_file(k)
        first_seg = 0
        if self.carry is not None:
            if keys[0] == self.carry[0]:
                self._extend_carry(meta, positions, bases, rles, int(seg_first[0]), int(seg_end[0]))
                first_seg = 1
                if len(keys) > 1:
                    self._close_carry()
            else:
                self._close_carry()
        if first_seg < len(keys):
            # the last run may continue in the next call: it becomes the carry; the others are complete
            last = len(keys) - 1
            self._decode_segments(keys[first_seg:last], seg_first[first_seg:last], seg_end[first_seg:last],
                                  meta, positions, bases, rles)
            self.carry = [keys[last], []]
            self._extend_carry(meta, positions, bases, rles, int(seg_first[last]), int(seg_end[last]))
        self.seconds["decode"] += time.time() - t0

    def _extend_carry(self, meta, positions, bases, rles, lo, hi):
        held = self.carry[1]
        if held is None:
            return
        if len(held) + hi - lo > _CARRY_LIMIT:
            self.carry[1] = None                      # too large to hold: from the file
            return
        for i in range(lo, hi):
            held.append((int(meta[i, 2]), positions[i].copy(), bases[i].copy(), rles[i].copy()))

    def _close_carry(self):
        key, held = self.carry
        self.carry = None
        if held is None or key in self.regions:       # held back too much, or the region was seen before: from the file
            if self.regions.get(key, b"") is not None:
                self._from_file(key)
            return
        ids = [h[0] for h in held]
        order = _string_order_once(ids, self.written.setdefault(key, set()))
        pos = np.ascontiguousarray(np.stack([held[i][1] for i in order]))
        b = np.ascontiguousarray(np.stack([held[i][2] for i in order]))
        r = np.ascontiguousarray(np.stack([held[i][3] for i in order]))
        blob, off = native_io.decode_regions(np.array([0, len(order)], np.int32), np.arange(len(order), dtype=np.int32),
                                             pos, b, r, 1)
        self._accept([key], blob, off)

    def _from_file(self, key):
        self.regions[key] = None
        self.seconds["from_file"] += 1
        if self.export is not None:
            self.export.from_file(key)

    def _decode_segments(self, keys, seg_first, seg_end, meta, positions, bases, rles):
        if not keys:
            return
        firsts, rows, good = [0], [], []
        ids = meta[:, 2]
        for key, a, e in zip(keys, seg_first.tolist(), seg_end.tolist()):
            if key in self.regions:                  # images of a region that was closed earlier: not back to back
                if self.regions[key] is not None:
                    self._from_file(key)
                continue
            seen = self.written.setdefault(key, set())
            seg_ids = ids[a:e].tolist()
            if not seen and all(0 <= x < 10 for x in seg_ids) and all(x < y for x, y in zip(seg_ids, seg_ids[1:])):
                order = range(e - a)                 # single digits in increasing order: string order as they stand
                seen.update(seg_ids)
            else:
                order = _string_order_once(seg_ids, seen)
            rows.extend(a + i for i in order)
            firsts.append(len(rows))
            good.append(key)
        if not good:
            return
        blob, off = native_io.decode_regions(np.array(firsts, np.int32), np.array(rows, np.int32), positions, bases, rles,
                                             self.decode_threads)
        self._accept(good, blob, off)

    def _accept(self, keys, blob, off):
        """Sequences of freshly decoded regions (in stream order): keep them, and hand the joins between stream
        neighbours that stitch will most likely make to the alignment workers."""
        blob = blob.tobytes()
        off = off.tolist()
        self.accept_sequences(keys, [blob[off[k]:off[k + 1]] for k in range(len(keys))])

    def accept_sequences(self, keys, seqs):
        """Regions with their sequences (bytes), in stream order (also the entry of a collector process, which receives
        them from the ranks: helen_amd.stitch_collect)."""
        if self.export is not None:
            self.export.write(keys, seqs)
            for key in keys:
                self.regions[key] = b""          # seen, and not with this process
            self.seconds["regions"] += len(keys)
            return
        jobs = []            # (left string, right string, the pair)
        regions, placed_of, rate = self.regions, self.placed, StitchOptions.BASE_ERROR_RATE

        def join(contig, a, b):
            # stitch joins the regions of a contig in (start, end) order; what it aligns is the last / first
            # `end(a) - start(b)` bases of the two (Stitch.py:141-147)
            if b[0] < a[1]:
                ov = a[1] - b[0]
                ov += int(ov * rate)
                sa, sb = regions.get((contig,) + a), regions.get((contig,) + b)
                if sa and sb:
                    jobs.append((sa[-ov:], sb[:ov], (contig,) + a + b))

        for key, seq in zip(keys, seqs):
            regions[key] = seq
            contig, span = key[0], key[1:]
            placed = placed_of.get(contig)
            if placed is None:
                placed = placed_of[contig] = []
            # the region's neighbours by POSITION among the regions decoded so far (images come in name order, which is
            # position order only between starts of the same number of digits)
            if not placed or placed[-1] < span:          # the ordinary case: the next region along the contig
                if placed:
                    a = placed[-1]
                    if span[0] < a[1] and seq:
                        sa = regions.get((contig,) + a)
                        if sa:
                            ov = a[1] - span[0]
                            ov += int(ov * rate)
                            jobs.append((sa[-ov:], seq[:ov], (contig,) + a + span))
                placed.append(span)
                continue
            i = bisect.bisect_left(placed, span)
            placed.insert(i, span)
            if i > 0:
                join(contig, placed[i - 1], span)
            if i + 1 < len(placed):
                join(contig, span, placed[i + 1])
        self.seconds["regions"] += len(keys)
        if jobs and self.pool is not None:
            self.seconds["joins_submitted"] += len(jobs)
            per = max(8, -(-len(jobs) // self.threads))
            for lo in range(0, len(jobs), per):
                self.pending.append(self.pool.submit(_align_jobs, jobs[lo:lo + per]))

    # ---- the end of the stream ----
    def abort(self):
        """The run failed: drop what is queued, let the workers go."""
        if self.export is not None:
            self.export.abandon()
            self.export = None
        if self.pool is not None:
            self.pool.shutdown(wait=False, cancel_futures=True)
            self.pool = None
        self.pending = []

    def finish(self):
        """-> StreamResult: every region of this file with its sequence (None = read it from the file) and the table of
        speculated joins.  Waits for the alignment workers."""
        if self.carry is not None:
            self._close_carry()
        for fut in self.pending:
            for left, right, res, pair in fut.result():
                self.joins[(left, right)] = res
                self.pair_joins[pair] = res
        self.pending = []
        if self.pool is not None:
            self.pool.shutdown()
            self.pool = None
        if self.export is not None:
            self.export.close()                  # the end marker: the collectors may finish
            self.export = None
            return StreamResult(self.file, {}, {}, dict(self.seconds, exported=True), {})
        return StreamResult(self.file, self.regions, self.joins, dict(self.seconds), self.pair_joins)


def _string_order_once(ids, seen):
    """Indices of `ids` in the STRING order of the chunk ids (sorted(set of str), Stitch.py:208-211), each id once -- the
    first image of an id is the one the writer stored (DataStore.py:123), also across calls (`seen`)."""
    first = {}
    for i, x in enumerate(ids):
        if x not in seen and x not in first:
            first[x] = i
    seen.update(first)
    return [first[x] for x in sorted(first, key=str)]


def _align_jobs(jobs):
    """[(left, right, pair)] -> [(left, right, (score, pos_a, pos_b), pair)] through helen_ssw_join_batch (one native call)."""
    blob = b"".join(x for job in jobs for x in job[:2])
    l_off, l_len, r_off, r_len = [], [], [], []
    at = 0
    for left, right, _ in jobs:
        l_off.append(at)
        l_len.append(len(left))
        at += len(left)
        r_off.append(at)
        r_len.append(len(right))
        at += len(right)
    out = native_io.ssw_join_batch(blob, l_off, l_len, r_off, r_len, StitchOptions.MATCH_PENALTY,
                                   StitchOptions.MISMATCH_PENALTY, StitchOptions.GAP_PENALTY,
                                   StitchOptions.GAP_EXTEND_PENALTY, StitchOptions.OVERLAP_THRESHOLD).tolist()
    return [(left, right, tuple(res), pair) for (left, right, pair), res in zip(jobs, out)]


class StreamResult(object):
    """What one rank's RegionStream hands the final pass; travels between processes as a file (save / load)."""

    def __init__(self, file, regions, joins, stats, pair_joins=None):
        self.file, self.regions, self.joins, self.stats = file, regions, joins, stats
        # the joins by region pair.  A pair's result was computed from the two regions' sequences as they are in
        # `regions` (a sequence never changes once it is there), so it is valid for exactly those
        self.pair_joins = {} if pair_joins is None else pair_joins

    def save(self, directory=None):
        fd, path = tempfile.mkstemp(prefix="helen_stream_%d_" % os.getpid(), suffix=".pkl", dir=directory)
        with os.fdopen(fd, "wb") as f:
            pickle.dump((self.file, self.regions, self.joins, self.stats, self.pair_joins), f,
                        protocol=pickle.HIGHEST_PROTOCOL)
        return path

    @staticmethod
    def load(path, remove=True):
        try:
            with open(path, "rb") as f:
                out = StreamResult(*pickle.load(f))
        finally:
            if remove:
                try:
                    os.unlink(path)
                except OSError:
                    pass
        return out


SPILL_FLOOR_BYTES = 1 << 30     # a RAM-backed directory that may not take this much is not used (Docker's default /dev/shm: 64 MB)


def spill_directory(expected_bytes=0):
    """Where a run parks region sequences between its processes: /dev/shm when it may take what is expected -- its free
    space AND half of the RAM the process tree may still claim (helen_amd.host_plan.ram_backed_budget_bytes: tmpfs pages are
    RAM) -- with a floor of 1 GiB; None otherwise (the caller uses the prediction directory)."""
    from .host_plan import ram_backed_budget_bytes
    for d in ("/dev/shm",):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            if ram_backed_budget_bytes(d) >= max(SPILL_FLOOR_BYTES, int(1.2 * expected_bytes)):
                return d
    return None


def _joined_from_slices(contig, rows, pair_joins):
    """The sequence of a contig as SLICES of its regions' sequences, when every join of the contig is an ordinary one
    whose result is in the table -- or None, and the caller goes through the reference's procedure join by join.

    `rows` = [(path, name, start, end, sequence)] in (start, end) order.  alignment_stitch (Stitch.py:96-190) keeps, of
    a join with alignment positions (pos_a, pos_b) over an overlap of `ov` bases, the running sequence up to
    len - ov + pos_a and the next sequence from pos_b on.  If the last `ov` bases of the running sequence are the left
    region's own (it kept at least that much after ITS left join) and the first `ov` bases of whatever the right
    region heads (the region itself, or the partial sequence of the run it starts: Stitch.py:257-301 stitches runs of
    regions first and the runs to each other afterwards) are the right region's own, then the two strings aligned are
    the two regions' tail and head -- the pair the stream has aligned already -- whatever the runs are, and the contig is
    region 0 [0 : e0] + region 1 [s1 : e1] + ...  Both conditions are checked for every region; gaps between
    neighbours (the ten-N filler and its warning) are taken along; anything else -- a region without a sequence, a pair
    that was never aligned, an alignment without a score or without an anchor, sequences of ten bases or fewer, two
    regions of one span -- leaves the contig to the general procedure."""
    m = len(rows)
    rate = StitchOptions.BASE_ERROR_RATE
    seqs = [e[4] for e in rows]
    if None in seqs:
        return None
    lens = [len(q) for q in seqs]
    if min(lens) <= 10:
        return None
    starts = [e[2] for e in rows]
    ends = [e[3] for e in rows]
    start_cut = [0] * m
    end_cut = list(lens)
    need_tail = [0] * m           # bases of its own that region i must still hold at its end (the overlap of its right join)
    need_head = [0] * m           # ... and at its start (the overlap of its left join)
    gaps = []
    get = pair_joins.get
    i = 0
    for a0, a1, b0, b1, n in zip(starts, ends, starts[1:], ends[1:], lens):
        if b0 < a1:
            res = get((contig, a0, a1, b0, b1))
            if res is None or res[0] == 0 or res[1] < 0 or res[2] < 0 or (a0 == b0 and a1 == b1):
                return None
            ov = a1 - b0
            ov += int(ov * rate)
            end_cut[i] = n - ov + res[1]
            start_cut[i + 1] = res[2]
            need_tail[i] = ov
            need_head[i + 1] = ov
        else:
            gaps.append(i + 1)
        i += 1
    for n, s0, e0, t, h in zip(lens, start_cut, end_cut, need_tail, need_head):
        if s0 > n - t or e0 < h or e0 <= s0:
            return None
    pieces = [memoryview(q)[s0:e0] for q, s0, e0 in zip(seqs, start_cut, end_cut)]
    total = sum(end_cut) - sum(start_cut) + 10 * len(gaps)
    warnings = []
    fill = b'N' * 10
    for g in reversed(gaps):
        pieces.insert(g, fill)
    for g in gaps:
        warnings.append("WARNING: NO OVERLAP IN CHUNKS:  " + str(contig) + " " + str(starts[g]) + " " + str(ends[g - 1]) + "\n")
    return pieces, total, warnings


def _align_unseen_neighbours(per_contig, joins, pair_joins, threads):
    """Neighbours that no stream has seen side by side -- the regions of a contig whose images were dealt to several
    ranks (MarginPolish writes a contig's regions into whichever thread's file; CallConsensusInterface.py:138-145 deals
    the files round-robin) -- are aligned here, all of them at once on `threads` worker threads, BEFORE the serial join
    pass, instead of one by one inside it.  `per_contig` = {contig: rows in (start, end) order}; fills `joins` (by the two
    strings) and `pair_joins` (by the two spans).  -> number of alignments made."""
    rate = StitchOptions.BASE_ERROR_RATE
    jobs = []
    for contig, rows in per_contig.items():
        for (_, _, a0, a1, sa), (_, _, b0, b1, sb) in zip(rows, rows[1:]):
            if b0 < a1 and sa and sb:
                pair = (contig, a0, a1, b0, b1)
                if pair_joins is not None and pair in pair_joins:
                    continue
                ov = a1 - b0
                ov += int(ov * rate)
                left, right = sa[-ov:], sb[:ov]
                known = joins.get((left, right))
                if known is None:
                    jobs.append((left, right, pair))
                elif pair_joins is not None and (a0, a1) != (b0, b1):
                    pair_joins[pair] = known
    if not jobs or not native_io.available():
        return 0
    per = max(8, -(-len(jobs) // (4 * max(1, threads))))
    with concurrent.futures.ThreadPoolExecutor(max(1, threads)) as pool:
        for done in pool.map(_align_jobs, [jobs[lo:lo + per] for lo in range(0, len(jobs), per)]):
            for left, right, res, pair in done:
                joins[(left, right)] = res
                if pair_joins is not None and pair[1:3] != pair[3:5]:
                    pair_joins[pair] = res
    return len(jobs)


def assemble_contigs(files, by_file, joins, pair_joins, threads, emit, fast=True, run_threads=None, quiet=False):
    """The join pass shared by finish_stitch and the collectors of a multi-rank run (helen_amd.stitch_collect): every
    contig of the streams `by_file` = {absolute path of a prediction file: its regions {(contig, start, end): sequence
    or None}}, `files` = those paths in the directory's listing order, in sorted contig order, handed to
    `emit(contig, pieces)` as a list of bytes-like pieces (header line and newline included; nothing for an empty
    sequence).  `threads` worker threads align what no stream has aligned; `run_threads` (default `threads`) is the
    thread count of the reference's run formula (Stitch.py:268-270: a contig is stitched in runs of
    max(2, regions // threads + 1) regions).  -> statistics."""
    run_threads = threads if run_threads is None else run_threads
    if not fast:
        pair_joins = {}
    # regions per contig: files in the directory's listing order, then (start, end) -- the order perform_stitch gives a
    # contig's regions (StitchInterface.py:84-95 lists a file's regions by name, Stitch.py:262 sorts by (start, end), the
    # sort is stable, and one file holds a span once: only the order of the FILES decides between equal spans)
    per_contig = {}
    for path in files:
        for (contig, start, end), seq in by_file[os.path.abspath(path)].items():
            per_contig.setdefault(contig, []).append((path, None, start, end, seq))
    by_span = operator.itemgetter(2, 3)
    for rows in per_contig.values():
        rows.sort(key=by_span)
    late = _align_unseen_neighbours(per_contig, joins, pair_joins if fast else None, threads)
    contigs = sorted(per_contig)
    hits = [0, 0]
    sliced = 0

    def aligner(left, right):
        got = joins.get((left, right))
        hits[0 if got is not None else 1] += 1
        return got

    def say(text):
        if not quiet:
            sys.stderr.write(text)

    for i, contig in enumerate(contigs):
        prefix = "{:04d}/{:04d}:".format(i, len(contigs))
        say("INFO: " + prefix + " PROCESSING CONTIG: " + contig + "\n")
        key_list = per_contig[contig]
        quick = _joined_from_slices(contig, key_list, pair_joins) if pair_joins else None
        if quick is not None:
            pieces, total, warnings = quick
            for w in warnings:
                sys.stderr.write(w)
            hits[0] += len(key_list) - 1 - len(warnings)
            sliced += 1
            say("INFO: " + prefix + " FINISHED PROCESSING " + contig + ", POLISHED SEQUENCE LENGTH: " + str(total) + ".\n")
            if total > 0:
                emit(contig, [b'>' + contig.encode() + b"\n"] + pieces + [b"\n"])
            continue
        n = max(StitchOptions.MIN_SEQUENCE_REQUIRED_FOR_MULTITHREADING, int(len(key_list) / run_threads) + 1)
        partial = []
        for lo in range(0, len(key_list), n):                       # FileManager.chunks
            chunk = []
            for path, name, start, end, seq in key_list[lo:lo + n]:
                if seq is None:
                    seq = native_io.region_sequence(path, contig, "%s-%d-%d" % (contig, start, end), as_bytes=True)
                chunk.append((contig, start, end, seq))
            chunk.sort(key=lambda e: (e[1], e[2]))
            c, s, e, running = _alignment_stitch(chunk, aligner)
            partial.append((c, s, e, bytes(running)))
        partial.sort(key=lambda e: (e[1], e[2]))
        sequence = _alignment_stitch(partial, aligner)[3] if partial else b""
        say("INFO: " + prefix + " FINISHED PROCESSING " + contig + ", POLISHED SEQUENCE LENGTH: " + str(len(sequence)) + ".\n")
        if len(sequence) > 0:
            emit(contig, [b'>' + contig.encode() + b"\n", sequence, b"\n"])
    answered, passed_on = native_io.ssw_fast_path_counts() if native_io.available() else (0, 0)
    return {"from_table": hits[0], "aligned_now": hits[1], "sliced": sliced, "contigs": len(contigs), "late": late,
            # this PROCESS's alignments so far (the stream's speculation included): answered by the aligner's exact-overlap
            # shortcut / taken through the three passes
            "shortcut": answered, "three_passes": passed_on}


def report_line(stats, from_file, threads):
    return ("INFO: STITCH PIPELINED BEHIND INFERENCE: %d JOINS FROM THE TABLE, %d ALIGNED NOW, %d REGION(S) READ BACK "
            "FROM THE FILES; %d OF %d CONTIG(S) ASSEMBLED FROM REGION SLICES%s.\n"
            % (stats["from_table"], stats["aligned_now"], from_file, stats["sliced"], stats["contigs"],
               ("" if not stats["late"] else "; %d JOIN(S) BETWEEN REGIONS OF DIFFERENT STREAMS ALIGNED ON %d THREAD(S) FIRST"
                % (stats["late"], threads))
               + ("" if not stats.get("shortcut", 0) + stats.get("three_passes", 0) else
                  "; %d OF %d ALIGNMENT(S) ANSWERED BY THE EXACT-OVERLAP SHORTCUT"
                  % (stats["shortcut"], stats["shortcut"] + stats["three_passes"]))))


class FastaWriter(object):
    """The FASTA written by a thread of its own (a contig is tens of megabytes; the write releases the interpreter lock)
    while the next contig is being joined.  put(pieces); close() joins the thread and raises what it met."""

    def __init__(self, path):
        import queue
        import threading
        self.q = queue.Queue(maxsize=4)
        self.error = []
        self.file = open(path, 'wb')
        self.offset = 0
        self.thread = threading.Thread(target=self._loop, daemon=True)
        self.thread.start()

    def _loop(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.error:
                continue
            try:
                self.file.write(b"".join(item))
            except BaseException as e:          # noqa: BLE001 -- raised by close()
                self.error.append(e)

    def put(self, pieces):
        self.q.put(pieces)

    def close(self):
        self.q.put(None)
        self.thread.join()
        self.file.close()
        if self.error:
            raise self.error[0]


def finish_stitch(results, input_directory, output_path, output_prefix, threads, fast=True):
    """perform_stitch (StitchInterface.py:40-106) from the streams of the run that has just written `input_directory`:
    the same contig order, region order, runs and joins -- and the same FASTA -- with the regions' sequences and most of
    the overlap alignments already there.  `results` = one StreamResult per prediction file of the directory."""
    by_file = {os.path.abspath(r.file): r for r in results}
    files = get_file_paths_from_directory(input_directory)
    missing = [f for f in files if os.path.abspath(f) not in by_file]
    if missing or len(files) != len(by_file):
        raise RuntimeError("the prediction directory and the streamed regions do not match (%s): run `helen stitch` on %s"
                           % (missing[:2], input_directory))
    joins, pair_joins = {}, {}
    for r in results:
        joins.update(r.joins)
        if fast:
            pair_joins.update(getattr(r, "pair_joins", None) or {})
    output_dir = file_manager.handle_output_directory(output_path)
    output_filename = os.path.join(output_dir, output_prefix + '.fa')
    sys.stderr.write("INFO: OUTPUT FILE: " + output_filename + "\n")
    fasta = FastaWriter(output_filename)
    import gc
    was_enabled = gc.isenabled()
    gc.disable()        # hundreds of thousands of tuples and slices, no cycles: generation scans would be a third of the pass
    try:
        stats = assemble_contigs(files, {k: r.regions for k, r in by_file.items()}, joins, pair_joins, threads,
                                 lambda contig, pieces: fasta.put(pieces), fast=fast)
    finally:
        fasta.close()
        if was_enabled:
            gc.enable()
    sys.stderr.write(report_line(stats, sum(r.stats.get("from_file", 0) for r in results), threads))
    return output_filename
