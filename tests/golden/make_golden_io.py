"""TEST INFRASTRUCTURE -- generates tests/golden/io_ref.json.gz by RUNNING the reference's own reader and writer.

Runs only in the build container (needs /root/reference; see tests/golden/reference_env.py for what the image lacks and
how it is supplied: h5py's few calls are spelled on top of libhdf5, torchvision.transforms is two empty callables):

    python tests/golden/make_golden_io.py

  * READER: the reference's `SequenceDataset` (helen/modules/python/models/dataloader_predict.py:18-95) is imported from
    /root/reference and its `__init__` / `__getitem__` are run over image files written by `io_case_files()` below
    (MarginPolish's schema as that reader expects it: `contig` a one-element string array, `contig_start` / `contig_end`
    / `feature_chunk_idx` one-element integer arrays of several widths, images of 1000, 613 and 1 rows, positions of
    several integer types).  The fixture stores, per image in the reference's own order, the 7-tuple it returned
    (arrays as SHA-1 of their bytes + dtype + shape).
  * WRITER: the reference's `DataStore.write_prediction` (helen/modules/python/DataStore.py:83-133) is run on seeded
    windows (several contigs, several chunk ids per region, a repeated window, padding rows) the way predict_gpu.py:176-179
    calls it; the fixture stores the resulting file as a tree {dataset path: dtype, shape, SHA-1}.
The tests regenerate the same inputs from the same seeds (io_case_files / writer_case are imported from this file, they
touch nothing of the reference) and compare what THIS package reads / writes with the fixture.
"""
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reference_env import ROOT, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "io_ref.json.gz")


def digest(a):
    a = np.ascontiguousarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape), "sha1": hashlib.sha1(a.tobytes()).hexdigest()}


def io_case_files(directory):
    """Two image files with every variant of the schema the reference's reader takes.  -> sorted paths"""
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    from helen_amd.weights import make_images
    img = make_images(9, seed=31)
    rng = np.random.default_rng(32)
    lengths = [1000, 613, 1000, 1, 999, 1000, 1000, 37, 1000]
    ints = [np.int64, np.int32, np.uint32, np.int64, np.uint16, np.int64, np.int16, np.uint64, np.int64]
    post = [np.int64, np.int32, np.uint32, np.int64, np.int64, np.uint16, np.int64, np.int32, np.int64]
    paths = []
    for k, lo, hi in ((0, 0, 5), (1, 5, 9)):
        path = os.path.join(directory, "images_%d.h5" % k)
        with hdf5.File(path, "w") as f:
            for i in range(lo, hi):
                L = lengths[i]
                start = 800 * (i // 2)
                contig = ["chr20", "contig_with_a_long_name.1", "c'q"][i % 3]
                base = "images/%s-%d-%d-%d/" % (contig.replace("'", ""), start, start + 1000, i % 2)
                f.write(base + "contig", contig)                                   # one-element fixed string array
                f.write(base + "contig_start", np.array([start], ints[i]))
                f.write(base + "contig_end", np.array([start + 1000], ints[i]))
                f.write(base + "feature_chunk_idx", np.array([i % 2], ints[i]))
                f.write(base + "image", img[i, :L], np.uint8)
                pos = np.stack([start + np.arange(L), rng.integers(0, 3, L), rng.integers(0, 2, L)], 1)
                f.write(base + "position", pos.astype(post[i]), post[i])
        paths.append(path)
    return paths


def writer_case():
    """Seeded windows for the writer: list of (contig, contig_start, contig_end, chunk_id, position [1000,3] int64,
    bases [1000], rles [1000]) in call order, with one window repeated (the writer skips it)."""
    rng = np.random.default_rng(33)
    windows = []
    for i in range(11):
        contig = ["ctgA", "ctgB", "z" * 40][i % 3]
        region = i // 4
        n = [1000, 1000, 613, 1000][i % 4]
        pos = np.full((1000, 3), -1, np.int64)
        pos[:n, 0] = 800 * region + np.arange(n)
        pos[:n, 1] = rng.integers(0, 3, n)
        pos[:n, 2] = rng.integers(0, 2, n)
        bases = np.zeros(1000, np.int64)
        rles = np.zeros(1000, np.int64)
        bases[:n] = rng.integers(0, 5, n)
        rles[:n] = rng.integers(0, 11, n)
        windows.append((contig, 800 * region, 800 * region + 1000, i % 4, pos, bases, rles))
    windows.append(windows[2])
    return windows


def walk(f, group="/", out=None):
    out = {} if out is None else out
    for k in f.keys(group):
        p = group.rstrip("/") + "/" + k
        try:
            f.keys(p)
            is_group = True
        except Exception:
            is_group = False
        if is_group and k not in ("position", "bases", "rles", "contig_start", "contig_end"):
            walk(f, p, out)
        else:
            out[p] = digest(f.read(p))
    return out


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference")
    install()
    sys.path.insert(0, "/root/reference")
    from helen.modules.python.DataStore import DataStore                       # the reference's own modules
    from helen.modules.python.models.dataloader_predict import SequenceDataset
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    d = tempfile.mkdtemp(prefix="helen_golden_io_")
    stderr, sys.stderr = sys.stderr, io.StringIO()
    try:
        paths = io_case_files(d)
        ds = SequenceDataset(None, file_list=paths)
        items = []
        for k in range(len(ds)):
            contig, start, end, chunk, image, position, path = ds[k]
            items.append({"file": os.path.basename(ds.all_images[k][0]), "name": ds.all_images[k][1], "contig": str(contig),
                          "contig_start": int(start), "contig_end": int(end), "chunk_id": int(chunk),
                          "image": digest(image), "position": digest(np.asarray(position, np.int64)),
                          "position_rows_before_padding": int((np.asarray(position)[:, 0] >= 0).sum()),
                          "returned_file": os.path.basename(path)})
        out = os.path.join(d, "pred.hdf")
        store = DataStore(out, mode="w")                                        # as predict_gpu.py:55 opens it
        for contig, start, end, chunk, pos, bases, rles in writer_case():
            store.write_prediction(contig, np.int64(start), np.int64(end), np.int64(chunk), pos, bases, rles, "x.h5")
        store.file_handler.close()
        with hdf5.File(out, "r") as f:
            tree = walk(f)
    finally:
        sys.stderr = stderr
        shutil.rmtree(d, ignore_errors=True)
    with io.TextIOWrapper(gzip.GzipFile(OUT, "wb", mtime=0)) as f:
        json.dump({"made_by": "tests/golden/make_golden_io.py (reference SequenceDataset and DataStore executed)",
                   "reader": items, "writer": tree}, f)
    print("wrote %s: %d reader items, %d datasets written, %d bytes" % (OUT, len(items), len(tree), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
