"""TEST INFRASTRUCTURE: a second, deliberately naive statement of the reference's stitch procedure
(helen/modules/python/Stitch.py:34-255 and StitchInterface.py:40-106), written independently of
helen_amd/stitch.py and of the native region decoder, to check those against.

Plain Python strings, dictionaries and per-character walks; nothing is optimised.  Alignments come from the
REFERENCE's own striped Smith-Waterman when oracle/_ref/libssw_ref.so is built (make -C oracle ref), else
from the product's aligner (which tests/test_stitch.py pins to the reference's cell for cell).
"""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SSW = os.path.join(ROOT, "oracle", "_ref", "libssw_ref.so")

MATCH, MISMATCH, GAP_OPEN, GAP_EXTEND = 4, 6, 8, 2     # Options.py:4-7
ANCHOR_RUN = 8                                         # Options.py: OVERLAP_THRESHOLD
LETTER = {0: "", 1: "A", 2: "C", 3: "G", 4: "T"}       # Options.py: label_decoder

_ref = None


def _align(left, right):
    """-> (score, reference_begin, cigar) of aligning `right` (query) to `left` (reference)."""
    global _ref
    if os.path.exists(REF_SSW):
        if _ref is None:
            _ref = ctypes.CDLL(REF_SSW)
        out = (ctypes.c_int * 6)()
        cig = ctypes.create_string_buffer(16 * (len(left) + len(right)) + 64)
        _ref.ssw_ref_align(left.encode(), len(left), right.encode(), MATCH, MISMATCH, GAP_OPEN, GAP_EXTEND, out,
                           cig, len(cig))
        return out[0], out[1], cig.value.decode()
    from helen_amd import native_io
    a = native_io.ssw_align(left, right, MATCH, MISMATCH, GAP_OPEN, GAP_EXTEND)
    return a.best_score, a.reference_begin, a.cigar_string


def anchor(reference_begin, cigar):
    """Walk the CIGAR one operation CHARACTER at a time; an anchor is the start of the first stretch of at least
    ANCHOR_RUN consecutive aligned columns ('=', 'X' or 'M').  -> (index in left, index in right) or (-1, -1)."""
    ops = []
    for count, op in re.findall(r"(\d+)(\D)", cigar):
        ops.extend(op * int(count))
    i_left, i_right = reference_begin, 0
    k = 0
    while k < len(ops):
        op = ops[k]
        if op in "=XM":
            run = 0
            while k + run < len(ops) and ops[k + run] in "=XM":
                run += 1
            if run >= ANCHOR_RUN:
                return i_left, i_right
            i_left += run
            i_right += run
            k += run
        elif op in "SI":
            i_right += 1
            k += 1
        elif op == "D":
            i_left += 1
            k += 1
        else:
            raise ValueError("unexpected CIGAR operation " + op)
    return -1, -1


def join(chunks):
    """chunks: [(contig, start, end, sequence)] -> (contig, start, end, sequence), Stitch.py:96-190."""
    chunks = sorted(chunks, key=lambda c: (c[1], c[2]))
    contig, run_start, run_end, run_seq = chunks[0]
    for _, start, end, seq in chunks[1:]:
        if start < run_end:
            n = run_end - start                      # BASE_ERROR_RATE is 0
            left = run_seq[-n:]                      # python: the whole string when n >= len
            right = seq[:n]
            score = _align(left, right) if left and right else (0, 0, "")
            if score[0] == 0:
                if len(right) > 10:
                    run_seq = run_seq + "N" * 10 + right
                    run_end = end
                continue
            a, b = anchor(score[1], score[2])
            if a == -1 or b == -1:
                if len(seq) > 10:
                    run_seq = run_seq[:-n] + left + "N" * 10 + seq
                    run_end = end
            else:
                run_seq = run_seq[:-n] + left[:a] + seq[b:]
                run_end = end
        else:
            if len(seq) > 10:
                run_seq = run_seq + "N" * 10 + seq
                run_end = end
    return contig, run_start, run_end, run_seq


def decode_region(f, contig, region):
    """One region's sequence from its images (Stitch.py:204-247): chunk ids in STRING order, the first image to
    mention a (pos, indx, split) key wins, keys with a negative pos or indx are skipped (the uint32-wrapped -1
    padding is NOT negative), keys in numeric order, base x run-length."""
    root = "predictions/%s/%s" % (contig, region)
    ids = sorted(k for k in f.keys(root) if k not in ("contig_start", "contig_end"))
    first = {}
    for cid in ids:
        pos = np.array(f.read(root + "/" + cid + "/position"), dtype=np.int64)
        bases = np.array(f.read(root + "/" + cid + "/bases"), dtype=np.int64)
        rles = np.array(f.read(root + "/" + cid + "/rles"), dtype=np.int64)
        for row, b, r in zip(pos, bases, rles):
            p, i, s = int(row[0]), int(row[1]), int(row[2])
            if i < 0 or p < 0:
                continue
            if (p, i, s) not in first:
                first[(p, i, s)] = (int(b), int(r))
    return "".join(LETTER[first[k][0]] * first[k][1] for k in sorted(first))


def stitch_directory(directory, threads=1):
    """{contig: sequence} for every `*hdf` file of a directory (StitchInterface.py:40-106 +
    Stitch.py:257-301): regions sorted by (start, end), cut into runs of max(2, n // threads + 1) regions that
    are joined separately, then the partial sequences are joined."""
    from helen_amd import hdf5
    files = [os.path.join(directory, n) for n in os.listdir(directory) if n[-3:] == "hdf"]
    contigs = {}
    for path in files:
        with hdf5.File(path, "r") as f:
            if "predictions" not in f:
                raise ValueError("no predictions in " + path)
            for contig in f.keys("predictions"):
                for region in f.keys("predictions/" + contig):
                    root = "predictions/%s/%s/" % (contig, region)
                    contigs.setdefault(contig, []).append(
                        (int(f.read(root + "contig_start")), int(f.read(root + "contig_end")), path, region))
    out = {}
    for contig, regions in contigs.items():
        regions.sort(key=lambda r: (r[0], r[1]))
        step = max(2, int(len(regions) / threads) + 1)
        partial = []
        for i in range(0, len(regions), step):
            seqs = []
            for start, end, path, region in regions[i:i + step]:
                with hdf5.File(path, "r") as f:
                    seqs.append((contig, start, end, decode_region(f, contig, region)))
            partial.append(join(seqs))
        out[contig] = join(partial)[3]
    return out
