"""stitch: prediction HDF5 files -> polished FASTA.

Same procedure and the same outputs as helen/modules/python/Stitch.py:14-301 and
StitchInterface.py:40-106:
  1. per region (`predictions/<contig>/<contig-start-end>`), the images' labels are merged by position
     key (first writer wins) and decoded base x run-length (`small_chunk_stitch`);
  2. neighbouring regions overlap (SEQ_OVERLAP): the tail of the running sequence and the head of the
     next one are aligned with striped Smith-Waterman (match 4, mismatch 6, gap 8/2, Options.py:4-7),
     the first match run of >= 8 is the anchor, and the sequences are joined there
     (`get_confident_positions`, `alignment_stitch`); 10 N's fill where no anchor exists.
What differs is mechanism only: the per-position merge and the aligner are native
(helen_amd/csrc/io.cpp, ssw.cpp -- the aligner reproduces the reference's SSW cell for cell), the
running sequence is a bytearray (the reference re-allocates the whole string at every join), and
worker processes are fed (file, region) lists.
"""
import concurrent.futures
import os
import re
import sys

from . import file_manager, hdf5, native_io


class StitchOptions(object):   # Options.py:1-10
    BASE_ERROR_RATE = 0.0
    label_decoder = {1: 'A', 2: 'C', 3: 'G', 4: 'T', 0: ''}
    MATCH_PENALTY = 4
    MISMATCH_PENALTY = 6
    GAP_PENALTY = 8
    GAP_EXTEND_PENALTY = 2
    MIN_SEQUENCE_REQUIRED_FOR_MULTITHREADING = 2
    OVERLAP_THRESHOLD = 8
    KMER_SIZE = 15


_CIGAR = re.compile(r'(\d+)(\w)')


def get_confident_positions(alignment):
    """(reference index, query index) of the first M run (= and X merged) of at least
    OVERLAP_THRESHOLD, or (-1, -1) (Stitch.py:34-94)."""
    cigar = alignment.cigar_string.replace('=', 'M').replace('X', 'M')
    grouped = []
    for length, op in _CIGAR.findall(cigar):
        if grouped and grouped[-1][0] == op:
            grouped[-1][1] += int(length)
        else:
            grouped.append([op, int(length)])
    ref_index = alignment.reference_begin
    read_index = 0
    for op, length in grouped:
        if op == 'M' and length >= StitchOptions.OVERLAP_THRESHOLD:
            return ref_index, read_index
        if op == 'S' or op == 'I':
            read_index += length
        elif op == 'D':
            ref_index += length
        elif op == 'M':
            ref_index += length
            read_index += length
        else:
            raise ValueError("ERROR: INVALID CIGAR OPERATION ENCOUNTERED WHILTE STITCHING: " + str(op) + "\n")
    return -1, -1


def alignment_stitch(sequence_chunks):
    """Join (contig, start, end, sequence) chunks in position order (Stitch.py:96-190)."""
    contig, running_start, running_end, running = _alignment_stitch(sequence_chunks)
    return contig, running_start, running_end, running.decode()


def _as_bytes(sequence):
    return sequence.encode() if isinstance(sequence, str) else sequence


def _ssw(left_chunk, right_chunk):
    return native_io.ssw_align(left_chunk, right_chunk, StitchOptions.MATCH_PENALTY, StitchOptions.MISMATCH_PENALTY,
                               StitchOptions.GAP_PENALTY, StitchOptions.GAP_EXTEND_PENALTY)


def _alignment_stitch(sequence_chunks, aligner=None):
    """alignment_stitch on bytes: the sequences of a contig are hundreds of megabytes, and every str <-> bytes
    conversion of the running sequence is a copy of all of it.  Chunk sequences may be str or bytes; the running
    sequence comes back as a bytearray.  `aligner(left, right)` may answer a join from a table of alignments made
    earlier -- (best score, pos_a, pos_b) for exactly these two strings, or None (helen_amd.stitch_stream)."""
    sequence_chunks = sorted(sequence_chunks, key=lambda e: (e[1], e[2]))
    contig, running_start, running_end, first = sequence_chunks[0]
    running = bytearray(_as_bytes(first))
    fill = b'N' * 10
    for i in range(1, len(sequence_chunks)):
        _, this_start, this_end, this_sequence = sequence_chunks[i]
        this_sequence = _as_bytes(this_sequence)
        if this_start < running_end:
            overlap_bases = running_end - this_start
            overlap_bases = overlap_bases + int(overlap_bases * StitchOptions.BASE_ERROR_RATE)
            # python slicing semantics of the reference: s[-n:] is all of s when n >= len(s)
            left_chunk = bytes(running[-overlap_bases:])
            right_chunk = bytes(this_sequence[:overlap_bases])
            alignment = known = None
            if len(left_chunk) > 0 and len(right_chunk) > 0:
                known = aligner(left_chunk, right_chunk) if aligner is not None else None
                if known is not None:
                    best_score = known[0]
                else:
                    alignment = _ssw(left_chunk, right_chunk)
                    best_score = alignment.best_score
            else:
                best_score = 0      # Align_cpp returns false on an empty sequence: score stays 0
            if best_score == 0:
                sys.stderr.write("WARNING: NO ALIGNMENT FOUND: " + str(this_start) + " " + str(this_end) + "\n")
                if len(right_chunk) > 10:
                    running += fill
                    running += right_chunk
                    running_end = this_end
            else:
                pos_a, pos_b = known[1:] if known is not None else get_confident_positions(alignment)
                if pos_a == -1 or pos_b == -1:
                    if alignment is None:
                        alignment = _ssw(left_chunk, right_chunk)      # (for the CIGAR text of the warning)
                    sys.stderr.write("WARNING: NO OVERLAPS IN ALIGNMENT : \n")
                    sys.stderr.write("LEFT : " + left_chunk.decode() + "\n")
                    sys.stderr.write("RIGHT: " + right_chunk.decode() + "\n")
                    sys.stderr.write("CIGAR: " + alignment.cigar_string + "\n")
                    if len(this_sequence) > 10:
                        # left_sequence + overlap_sequence is the running sequence itself
                        running += fill
                        running += this_sequence
                        running_end = this_end
                else:
                    # running[:-overlap] + left_chunk[:pos_a] + this[pos_b:]
                    keep = len(running) - len(left_chunk) + pos_a
                    del running[keep:]
                    running += memoryview(this_sequence)[pos_b:]
                    running_end = this_end
        else:
            sys.stderr.write("WARNING: NO OVERLAP IN CHUNKS:  " + str(contig) + " " + str(this_start) + " "
                             + str(running_end) + "\n")
            if len(this_sequence) > 10:
                running += fill
                running += this_sequence
                running_end = this_end
    return contig, running_start, running_end, running


def _region_sequence_py(file_name, contig, chunk_name):
    """Pure-Python fallback of native_io.region_sequence (no libhelen_io.so)."""
    with hdf5.File(file_name, "r") as f:
        root = "predictions/%s/%s" % (contig, chunk_name)
        chunks = sorted(set(f.keys(root)) - {"contig_start", "contig_end"})
        base_d, rle_d = {}, {}
        for chunk in chunks:
            bases = f.read(root + "/" + chunk + "/bases")
            rles = f.read(root + "/" + chunk + "/rles")
            positions = f.read(root + "/" + chunk + "/position", "int64")
            for (pos, indx, split), b, r in zip(positions.tolist(), bases.tolist(), rles.tolist()):
                if indx < 0 or pos < 0:
                    continue
                if (pos, indx, split) not in base_d:
                    base_d[(pos, indx, split)] = b
                    rle_d[(pos, indx, split)] = r
        return ''.join(StitchOptions.label_decoder[base_d[k]] * int(rle_d[k]) for k in sorted(base_d))


def small_chunk_stitch(contig, small_chunk_keys):
    """Decode every region of `small_chunk_keys` = [(contig, file, region name, start, end)] and stitch
    them (Stitch.py:192-255)."""
    name_sequence_tuples = []
    for contig_name, file_name, chunk_name, contig_start, contig_end in small_chunk_keys:
        if native_io.available():
            sequence = native_io.region_sequence(file_name, contig, chunk_name, as_bytes=True)
        else:
            sequence = _region_sequence_py(file_name, contig, chunk_name)
        name_sequence_tuples.append((contig, contig_start, contig_end, sequence))
    name_sequence_tuples = sorted(name_sequence_tuples, key=lambda e: (e[1], e[2]))
    contig, start, end, running = _alignment_stitch(name_sequence_tuples)
    return contig, start, end, bytes(running)


_SPILL_BYTES = 1 << 20


def _small_chunk_stitch_worker(contig, small_chunk_keys, spill_dir):
    """small_chunk_stitch in a worker process: a long sequence goes back through a file in `spill_dir` (RAM-backed when
    there is one) instead of the result pipe -- a pickle through a pipe moves about 1 GB/s, and a contig's runs are
    tens of megabytes each, collected by one thread of the parent."""
    contig, start, end, sequence = small_chunk_stitch(contig, small_chunk_keys)
    if spill_dir is None or len(sequence) < _SPILL_BYTES:
        return contig, start, end, sequence
    import tempfile
    path = None
    try:
        fd, path = tempfile.mkstemp(prefix="helen_stitch_%d_" % os.getppid(), suffix=".seq", dir=spill_dir)
        with os.fdopen(fd, "wb") as f:
            f.write(sequence)
    except OSError:                      # no room there: the pipe it is
        if path is not None:
            try:
                os.unlink(path)
            except OSError:
                pass
        return contig, start, end, sequence
    return contig, start, end, _Spilled(path)


class _Spilled(object):
    def __init__(self, path):
        self.path = path

    def take(self):
        try:
            with open(self.path, "rb") as f:
                return f.read()
        finally:
            try:
                os.unlink(self.path)
            except OSError:
                pass


def _spill_dir():
    if os.environ.get("HELEN_STITCH_SPILL", "1") == "0":
        return None
    for d in ("/dev/shm",):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            return d
    return None


# exceptions of worker runs of the LAST perform_stitch of this process (the reference only prints them); every call
# collects its own list and leaves it here when it ends -- concurrent calls do not share state
FAILED_RUNS = []


def _submit_contig(contig, sequence_chunk_keys, threads, executor):
    """First half of create_consensus_sequence: sort the regions, cut them into runs (FileManager.chunks), hand the
    runs to the pool.  -> list of futures / finished (contig, start, end, sequence) tuples, in any order."""
    key_list = sorted(((contig, f, key, int(st), int(end)) for f, key, st, end in sequence_chunk_keys),
                      key=lambda e: (e[3], e[4]))
    if not key_list:
        return []
    n = max(StitchOptions.MIN_SEQUENCE_REQUIRED_FOR_MULTITHREADING, int(len(key_list) / threads) + 1)
    file_chunks = [key_list[i:i + n] for i in range(0, len(key_list), n)]   # FileManager.chunks
    if executor is None:
        return [small_chunk_stitch(contig, fc) for fc in file_chunks]
    spill = _spill_dir()
    return [executor.submit(_small_chunk_stitch_worker, contig, fc, spill) for fc in file_chunks]


def _finish_contig(jobs, failed=None):
    """Second half: collect the runs' sequences and stitch them (bytes-like).  `failed` (a list) receives the text of
    every run that raised."""
    failed = FAILED_RUNS if failed is None else failed
    sequence_chunks = []
    for job in jobs:
        if isinstance(job, concurrent.futures.Future):
            if job.exception() is None:
                contig, start, end, sequence = job.result()
                if isinstance(sequence, _Spilled):
                    sequence = sequence.take()
                sequence_chunks.append((contig, start, end, sequence))
            else:
                # the reference prints the exception and stitches the contig from the runs that survived
                # (Stitch.py:283-291); a DEAD WORKER is different: the pool is broken for every contig still to come
                from concurrent.futures.process import BrokenProcessPool
                if isinstance(job.exception(), BrokenProcessPool):
                    raise RuntimeError("a stitch worker process died (%s): the FASTA would be truncated"
                                       % job.exception())
                sys.stderr.write("ERROR: " + str(job.exception()) + "\n")
                failed.append(str(job.exception()))
        else:
            sequence_chunks.append(job)
    if not sequence_chunks:
        return b""
    sequence_chunks = sorted(sequence_chunks, key=lambda e: (e[1], e[2]))
    return _alignment_stitch(sequence_chunks)[3]


def create_consensus_sequence(contig, sequence_chunk_keys, threads, executor=None):
    """(Stitch.py:257-301): sort the regions, stitch runs of them in worker processes, then stitch the
    partial sequences.  `executor`: a process pool to use (perform_stitch keeps one for all contigs; the
    reference starts a new one per contig)."""
    own = executor is None and threads > 1
    ex = _new_pool(threads) if own else executor
    try:
        return _finish_contig(_submit_contig(contig, sequence_chunk_keys, threads, ex if threads > 1 else None))
    finally:
        if own:
            ex.shutdown()


def _worker_ready():
    return native_io.available()


def _new_pool(threads):
    """A pool whose workers start (and import this module) right away, while the parent lists the regions."""
    import multiprocessing as mp
    pool = concurrent.futures.ProcessPoolExecutor(max_workers=threads, mp_context=mp.get_context("spawn"))
    for _ in range(threads):
        pool.submit(_worker_ready)
    return pool


def _regions_of(prediction_file, contig):
    """[(file, region name, contig_start, contig_end)] of one file for `contig`, regions in name order
    (StitchInterface.py:84-95) -- through the native lister when there is one: two dataset reads per region
    through a Python binding are minutes at 300 k regions."""
    if native_io.available():
        listed = native_io.list_regions(prediction_file, contig)
        return [] if listed is None else [(prediction_file, name, st, en) for name, st, en in listed]
    out = []
    with hdf5.File(prediction_file, "r") as f:
        if contig not in f.keys("predictions"):
            return out
        for chunk_key in sorted(f.keys("predictions/" + contig)):
            root = "predictions/%s/%s/" % (contig, chunk_key)
            out.append((prediction_file, chunk_key, int(f.read(root + "contig_start")), int(f.read(root + "contig_end"))))
    return out


def get_file_paths_from_directory(directory_path):
    """`*hdf` files of a directory (StitchInterface.py:30-37)."""
    return [os.path.abspath(os.path.join(directory_path, f)) for f in os.listdir(directory_path)
            if os.path.isfile(os.path.join(directory_path, f)) and f[-3:] == 'hdf']


def perform_stitch(input_directory, output_path, output_prefix, threads):
    """Every contig of every prediction file -> `<output_path>/<output_prefix>.fa`
    (StitchInterface.py:40-106).  Unlike the reference, which prints a failed run's exception and returns a FASTA
    stitched from what survived, this raises after writing that FASTA when any run failed -- the error names the
    incomplete file; $HELEN_STITCH_KEEP_GOING=1 restores the reference's behaviour (print, return the path)."""
    native_io.close_readers() if native_io.available() else None
    failed = []
    all_prediction_files = get_file_paths_from_directory(input_directory)
    all_contigs = set()
    contigs_of = {}
    for prediction_file in sorted(all_prediction_files):
        with hdf5.File(prediction_file, "r") as f:
            if "predictions" in f:
                contigs_of[prediction_file] = f.keys("predictions")
                all_contigs.update(contigs_of[prediction_file])
            else:
                raise ValueError("ERROR: INVALID HDF5 FILE, FILE DOES NOT CONTAIN predictions KEY.\n")
    # Region lists file by file (one file mapped at a time), not contig by contig across all files: an assembly is
    # thousands of contigs in a handful of files.  Per contig the files keep the order of all_prediction_files.
    regions_of = {contig: [] for contig in all_contigs}
    for prediction_file in all_prediction_files:
        for contig in contigs_of[prediction_file]:
            regions_of[contig].extend(_regions_of(prediction_file, contig))
    output_dir = file_manager.handle_output_directory(output_path)
    output_filename = os.path.join(output_dir, output_prefix + '.fa')
    sys.stderr.write("INFO: OUTPUT FILE: " + output_filename + "\n")
    executor = _new_pool(threads) if threads > 1 else None      # one pool for all contigs
    # Contigs are pipelined: the runs of the next contigs are already with the workers while the parent stitches
    # and writes the current one (an assembly is thousands of contigs, most of them a few regions long; the
    # reference finishes one contig -- and one process pool -- before it looks at the next).  FASTA order is the
    # contig order either way.
    contigs = sorted(all_contigs)
    look_ahead = 64 * max(1, threads)           # regions handed out and not yet collected
    pending, in_flight, nxt = [], 0, 0
    try:
        with open(output_filename, 'wb') as fasta:
            while nxt < len(contigs) or pending:
                while nxt < len(contigs) and (not pending or (executor is not None and in_flight < look_ahead)):
                    contig = contigs[nxt]
                    prefix = "{:04d}/{:04d}:".format(nxt, len(contigs))
                    sys.stderr.write("INFO: " + prefix + " PROCESSING CONTIG: " + contig + "\n")
                    chunk_name_tuple = regions_of.pop(contig)
                    pending.append((contig, prefix, len(chunk_name_tuple),
                                    _submit_contig(contig, chunk_name_tuple, threads, executor)))
                    in_flight += len(chunk_name_tuple)
                    nxt += 1
                contig, prefix, regions, jobs = pending.pop(0)
                consensus_sequence = _finish_contig(jobs, failed)
                in_flight -= regions
                sys.stderr.write("INFO: " + prefix + " FINISHED PROCESSING " + contig
                                 + ", POLISHED SEQUENCE LENGTH: " + str(len(consensus_sequence)) + ".\n")
                if consensus_sequence is not None and len(consensus_sequence) > 0:
                    fasta.write(b'>' + contig.encode() + b"\n")
                    fasta.write(consensus_sequence)
                    fasta.write(b"\n")
    finally:
        if executor is not None:
            executor.shutdown()
            spill = _spill_dir()      # whatever an interrupted run left behind
            if spill is not None:
                import glob
                for leftover in glob.glob(os.path.join(spill, "helen_stitch_%d_*.seq" % os.getpid())):
                    try:
                        os.unlink(leftover)
                    except OSError:
                        pass
    FAILED_RUNS[:] = failed
    if failed:
        text = ("%d stitch run(s) failed, %s is INCOMPLETE (stitched from the runs that survived); first error: %s"
                % (len(failed), output_filename, failed[0]))
        if os.environ.get("HELEN_STITCH_KEEP_GOING", "") == "1":      # StitchInterface.py:40-106 prints and carries on
            sys.stderr.write("ERROR: " + text + "\n")
            return output_filename
        raise RuntimeError(text)
    return output_filename
