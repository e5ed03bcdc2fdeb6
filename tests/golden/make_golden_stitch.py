"""TEST INFRASTRUCTURE -- generates tests/golden/stitch_ref.json.gz by RUNNING the reference's own stitch logic.

Runs only in the build container (needs /root/reference and oracle/_ref/libssw_ref.so, `make -C oracle ref`):

    python tests/golden/make_golden_stitch.py

What is executed is the reference's `Stitch.alignment_stitch` (helen/modules/python/Stitch.py:96-190: overlap
arithmetic, anchor search, the N x 10 fillers, the short-chunk and empty-region rules) and its
`Stitch.get_confident_positions` (Stitch.py:33-94), imported from /root/reference, on seeded chains of chunks and on
seeded CIGARs.  The fixture holds inputs and the reference's outputs (data only).

A second part runs the reference's `Stitch.small_chunk_stitch` (Stitch.py:192-255: the per-position dictionaries, first
writer wins over chunk ids in string order) and `Stitch.create_consensus_sequence` (Stitch.py:257-301, 1 and 3
workers) on prediction FILES written here by this package's DataStore from seeded regions (duplicate keys across
chunk ids, gaps, padding rows, noisy regions, holes); the fixture stores the regions' rows and the reference's results.

What the image lacks for importing and running the reference's module (its pybind11 aligner module `HELEN`, `h5py`,
`np.int`) is supplied by tests/golden/reference_env.py, which documents each item: the aligner is the reference's own
ssw.c compiled in place, h5py's few calls are spelled on top of libhdf5.
"""
import gzip
import io
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reference_env import REF_SSW, ROOT, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "stitch_ref.json.gz")


def region_rows(rng, truth_b, truth_r, p0, length, noisy):
    """Rows of one region as chunk id -> (positions [n,3], bases, rles): the region's positions cut into 1-3 chunk ids
    whose key ranges overlap (the later id, in STRING order, repeats keys with other labels: first writer wins),
    insert columns, gap labels, padding rows in the middle."""
    import numpy as np
    keys = []
    for p in range(p0, p0 + length):
        keys.append((p, 0, 0))
        if rng.random() < 0.15:
            keys.append((p, 1, rng.randrange(0, 2)))
    labels = {}
    for (p, i, sp) in keys:
        if noisy:
            labels[(p, i, sp)] = (rng.randrange(0, 5), rng.randrange(0, 4))
        elif i == 0:
            labels[(p, i, sp)] = (int(truth_b[p]), int(truth_r[p]))
        else:
            labels[(p, i, sp)] = (0, 0) if rng.random() < 0.7 else (rng.randrange(1, 5), 1)
    n_ids = rng.randrange(1, 4)
    ids = rng.sample([0, 1, 2, 10, 11], n_ids)
    chunks = {}
    for k, cid in enumerate(sorted(ids, key=str)):
        lo = 0 if k == 0 else max(0, len(keys) * k // n_ids - 8)
        hi = len(keys) if k == n_ids - 1 else len(keys) * (k + 1) // n_ids
        rows = keys[lo:hi]
        b = [labels[q][0] for q in rows]
        r = [labels[q][1] for q in rows]
        if k > 0:                                            # repeated keys come with other labels
            for t in range(min(8, len(rows))):
                b[t], r[t] = rng.randrange(0, 5), rng.randrange(0, 4)
        pos = [list(q) for q in rows]
        if rng.random() < 0.3:                               # padding rows in the middle of an image
            at = rng.randrange(0, len(pos))
            pos[at:at] = [[-1, -1, -1]] * 2
            b[at:at] = [rng.randrange(0, 5)] * 2
            r[at:at] = [rng.randrange(0, 4)] * 2
        chunks[str(cid)] = {"position": pos, "bases": b, "rles": r}
    return chunks


def write_case(directory, case):
    import numpy as np
    from helen_amd.data_store import DataStore
    stores = {}
    for region in case["regions"]:
        path = os.path.join(directory, region["file"])
        if path not in stores:
            stores[path] = DataStore(path, "w")
        for cid, rows in sorted(region["chunks"].items()):
            n = len(rows["bases"])
            P = np.full((1000, 3), -1, np.int64)
            B = np.zeros(1000, np.uint8)
            R = np.zeros(1000, np.uint8)
            P[:n], B[:n], R[:n] = np.array(rows["position"], np.int64).reshape(n, 3), rows["bases"], rows["rles"]
            stores[path].write_prediction(region["contig"], region["start"], region["end"], int(cid), P, B, R)
    for st in stores.values():
        st.close()
    return sorted(stores)


def directory_case(rng, second_contig=False):
    import numpy as np
    regions = []
    for contig in (("ctg", "a_second_contig") if second_contig else ("ctg",)):
        total = 700 if contig == "ctg" else 260
        truth_b = np.array([rng.randrange(1, 5) for _ in range(total)])
        truth_r = np.array([rng.randrange(1, 4) for _ in range(total)])
        p0, k, count = 0, 0, 0
        while p0 + 20 < total and count < 9:
            length = rng.choice([30, 60, 60, 90])
            length = min(length, total - p0)
            noisy = rng.random() < 0.12
            regions.append({"file": "p_%d.hdf" % (k % 2), "contig": contig, "start": p0, "end": p0 + length,
                            "chunks": region_rows(rng, truth_b, truth_r, p0, length, noisy)})
            step = rng.random()
            p0 = p0 + length - rng.choice([25, 25, 12]) if step < 0.8 else p0 + length + (0 if step < 0.9 else 7)
            k += 1
            count += 1
    return {"regions": regions}


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice("ACGT"))
            out.append(c)
        elif x < rate:
            out.append(rng.choice("ACGTN"))
        else:
            out.append(c)
    return "".join(out)


def chain(rng):
    """2-6 chunks cut from one truth sequence: mutated / unrelated / empty / tiny chunks, overlaps above and below the
    anchor run of 8, abutting, gapped and nested chunks, homopolymer-rich truth, shuffled order."""
    if rng.random() < 0.3:
        truth = "".join(rng.choice("ACGT") * rng.randrange(1, 7) for _ in range(rng.choice([20, 100, 300])))
    else:
        truth = rand_seq(rng, rng.choice([60, 300, 1200]))
    chunks, start = [], 0
    for _ in range(rng.randrange(2, 7)):
        length = rng.choice([3, 9, 10, 11, 40, 150, 400])
        end = min(len(truth), start + length)
        seq = truth[start:end]
        kind = rng.random()
        if kind < 0.35:
            seq = mutate(rng, seq, rng.choice([0.01, 0.05, 0.2, 0.5]))
        elif kind < 0.45:
            seq = rand_seq(rng, len(seq))
        elif kind < 0.5:
            seq = ""
        chunks.append(["ctg", start, end, seq])
        step = rng.random()
        if step < 0.55:
            start = max(start + 1, end - rng.choice([1, 3, 7, 8, 9, 20, 60, 200]))
        elif step < 0.7:
            start = end
        elif step < 0.85:
            start = end + rng.choice([1, 50])
        else:
            start = start + rng.choice([1, 5])
        if start >= len(truth):
            break
    rng.shuffle(chunks)
    return chunks


def main():
    if not os.path.isdir("/root/reference") or not os.path.exists(REF_SSW):
        sys.exit("needs /root/reference and oracle/_ref/libssw_ref.so (make -C oracle ref)")
    Alignment = install()
    sys.path.insert(0, "/root/reference")
    from helen.modules.python.Stitch import Stitch           # the reference's own module
    rng = random.Random(20260929)
    joins, anchors = [], []
    stderr, sys.stderr = sys.stderr, io.StringIO()            # the procedure warns on stderr
    try:
        while len(joins) < 400:
            chunks = chain(rng)
            if len(chunks) < 2:
                continue
            contig, start, end, seq = Stitch().alignment_stitch([tuple(c) for c in chunks])
            joins.append({"chunks": chunks, "result": [contig, start, end, seq]})
        for _ in range(1500):
            parts, last = [], None
            for _ in range(rng.randrange(1, 9)):
                op = rng.choice([o for o in "=XIDSM" if o != last or o in "=X"])
                parts.append("%d%s" % (rng.choice([1, 2, 3, 5, 7, 8, 9, 30]), op))
                last = op
            a = Alignment()
            a.cigar_string = "".join(parts)
            a.reference_begin = rng.randrange(0, 50)
            try:
                got = list(Stitch.get_confident_positions(a))
            except ValueError:
                got = "ValueError"
            anchors.append({"cigar": a.cigar_string, "reference_begin": a.reference_begin, "result": got})
    finally:
        sys.stderr = stderr
    import shutil
    import tempfile
    directories = []
    stderr, sys.stderr = sys.stderr, io.StringIO()
    try:
        from helen.modules.python.StitchInterface import perform_stitch       # the reference's own driver
        for n in range(14):
            case = directory_case(rng, second_contig=n % 3 == 0)
            d = tempfile.mkdtemp(prefix="helen_golden_")
            try:
                files = write_case(d, case)
                case["perform_stitch"] = {}
                for t in (1, 3):            # StitchInterface.py:40-106: every contig of every file -> FASTA
                    fasta = perform_stitch(d, os.path.join(d, "fasta%d" % t), "asm", t)
                    case["perform_stitch"][str(t)] = open(os.path.join(d, "fasta%d" % t, "asm.fa")).read()
                    assert fasta is None or True
                keys = sorted((("ctg", os.path.join(d, r["file"]), "ctg-%d-%d" % (r["start"], r["end"]), r["start"], r["end"])
                               for r in case["regions"] if r["contig"] == "ctg"), key=lambda e: (e[3], e[4]))
                run = Stitch().small_chunk_stitch("ctg", keys)
                case["small_chunk_stitch"] = [run[0], int(run[1]), int(run[2]), run[3]]
                tuples = [(k[1], k[2], k[3], k[4]) for k in keys]
                case["create_consensus_sequence"] = {str(t): Stitch().create_consensus_sequence("ctg", tuples, t)
                                                     for t in (1, 3)}
                case["files"] = [os.path.basename(f) for f in files]
                directories.append(case)
            finally:
                shutil.rmtree(d, ignore_errors=True)
    finally:
        sys.stderr = stderr
    with io.TextIOWrapper(gzip.GzipFile(OUT, "wb", mtime=0)) as f:      # mtime 0: the same bytes every time
        json.dump({"made_by": "tests/golden/make_golden_stitch.py (reference Stitch.py executed, reference ssw.c alignments)",
                   "joins": joins, "anchors": anchors, "directories": directories}, f)
    fillers = sum("N" * 10 in j["result"][3] for j in joins)
    print("wrote %s: %d joins (%d with fillers), %d anchor cases, %d directories (%d with fillers), %d bytes"
          % (OUT, len(joins), fillers, len(anchors), len(directories),
             sum("N" * 10 in d["small_chunk_stitch"][3] for d in directories), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
