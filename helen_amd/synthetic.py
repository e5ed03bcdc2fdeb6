"""Synthetic MarginPolish-style image files: what the tests, the CLI smoke run and the
end-to-end benchmark read.  Schema = what the reference reader expects
(helen/modules/python/models/dataloader_predict.py:64-70):
    images/<name>/{contig (str[1]), contig_start[1], contig_end[1], feature_chunk_idx[1],
                   image [l, 90] uint8, position [l, 3] int}
"""
import os

import numpy as np

from . import hdf5
from .options import ImageSizeOptions
from .weights import make_images


def write_image_file(path, images, contig="chr20_synth", first_window=0, lengths=None,
                     chunks_per_region=1, labels=None, gzip=None, chunks=None, shuffle=False, fletcher32=False, libver=None,
                     string="fixed"):
    """Write `images` (uint8 [n, 1000, 90]) as n images of one file.  Window k covers
    contig_start = 800*k .. +1000 (SEQ_OVERLAP 200, Options.py:17); `lengths[i] < 1000` stores a
    short image (the reader pads it).  labels = (label_base, label_run_length) uint8 [n, 1000] makes
    it a labeled file as the evaluation loader reads it (models/dataloader.py:59-61); gzip = 1..9 stores
    image and position chunked + deflated (as an h5py writer with compression="gzip" would); chunks = (rows, 90) stores them
    chunked (position in chunks of the same number of rows); shuffle / fletcher32 add those filters; libver="latest"
    writes the newest file format (superblock 3, version 2 object headers, dense groups, version 4 layouts)."""
    n = images.shape[0]
    packed = bool(gzip or shuffle or fletcher32)
    ich = tuple(chunks) if chunks is not None else ((256, 90) if packed else None)
    with hdf5.File(path, "w", libver=libver) as f:
        for i in range(n):
            k = first_window + i
            region = k // chunks_per_region
            chunk = k % chunks_per_region
            start = 800 * region
            L = int(lengths[i]) if lengths is not None else ImageSizeOptions.SEQ_LENGTH
            name = "%s-%d-%d-%d" % (contig, start, start + 1000, chunk)
            base = "images/" + name + "/"
            f.write(base + "contig", contig, string=string)
            f.write(base + "contig_start", np.array([start], np.int64))
            f.write(base + "contig_end", np.array([start + 1000], np.int64))
            f.write(base + "feature_chunk_idx", np.array([chunk], np.int64))
            f.write(base + "image", images[i, :L], np.uint8, chunks=ich, gzip=gzip, shuffle=shuffle, fletcher32=fletcher32)
            pos = np.zeros((L, 3), np.int64)
            pos[:, 0] = start + np.arange(L)
            pch = (ich[0], 3) if chunks is not None else ((L, 3) if packed else None)
            f.write(base + "position", pos, np.int64, chunks=pch, gzip=gzip, shuffle=shuffle, fletcher32=fletcher32)
            if labels is not None:
                f.write(base + "label_base", labels[0][i, :L], np.uint8)
                f.write(base + "label_run_length", labels[1][i, :L], np.uint8)


def write_image_file_direct(path, images, contig="chr20_synth", first_window=0, lengths=None):
    """The same file as write_image_file (one chunk id per region, no labels, contiguous datasets), written by
    the direct emitter of libhelen_io.so at ~100 k windows/s instead of ~1.3 k: what the end-to-end
    benchmarks use to make chr20-scale inputs in seconds.  (The reader tests keep libhdf5-written files.)"""
    from . import native_io
    n = images.shape[0]
    k = first_window + np.arange(n, dtype=np.int64)
    L = np.full(n, ImageSizeOptions.SEQ_LENGTH, np.int32) if lengths is None else np.asarray(lengths, np.int32)
    native_io.emit_images(path, contig, 800 * k, np.zeros(n, np.int64), L, images)


def write_image_dir(directory, n_windows, n_files=4, seed=20260928, mode="uniform", short_every=0, direct=False):
    """A directory of `n_files` image files holding `n_windows` windows in total.  Returns the
    list of files.  short_every > 0 makes every such window a short image (613 positions); direct=True
    writes through the emitter of libhelen_io.so (fast; same schema and values)."""
    os.makedirs(directory, exist_ok=True)
    per = (n_windows + n_files - 1) // n_files
    files = []
    done = 0
    for fi in range(n_files):
        n = min(per, n_windows - done)
        if n <= 0:
            break
        img = make_images(n, seed=seed + fi, mode=mode)
        lengths = None
        if short_every > 0:
            lengths = np.full(n, ImageSizeOptions.SEQ_LENGTH)
            lengths[short_every - 1::short_every] = 613
        path = os.path.join(directory, "synthetic_images_%03d.h5" % fi)
        (write_image_file_direct if direct else write_image_file)(path, img, first_window=done, lengths=lengths)
        files.append(path)
        done += n
    return files


# ---- a learnable synthetic polishing task: what tests/golden/make_trained_synth.py trains the reference model on ----
def make_pileup_task(n_windows, seed=20260928, coverage=14, base_error=0.12, rl_error=0.35):
    """Synthetic pileup windows WITH a ground truth, shaped like a run-length-compressed MarginPolish image: every
    position has a true base (label 0 = gap, 1..4 = A, C, G, T; gaps 10 % of the positions) and a true run length
    (0 for gaps, else 1..10, geometric), and `coverage` reads vote for what they saw -- the true base with probability
    1 - base_error (else another symbol), the true run length with probability 1 - rl_error (else one off: runs of
    five and more are under-called three times out of four, as nanopore reads do) -- on one of two strands.  Feature layout: strand * 45 + (base - 1) * 11 + run_length for a base, strand * 45 + 44 for a
    gap; a count of k votes is stored as min(255, 8 k).  Counts, labels and the noise process are all seeded numpy.
    -> (images uint8 [n, 1000, 90], label_base uint8 [n, 1000], label_rle uint8 [n, 1000])

    Nothing about real MarginPolish data is claimed: the point is a task on which the reference network can be
    TRAINED offline, so that parity and reduced-precision figures can be quoted on a network with trained-looking
    weights and confident outputs instead of random ones."""
    rng = np.random.default_rng(seed)
    L, F = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
    base = rng.integers(1, 5, size=(n_windows, L))
    base[rng.random((n_windows, L)) < 0.10] = 0
    rl = np.minimum(10, rng.geometric(0.55, size=(n_windows, L)))
    rl[base == 0] = 0
    votes = np.zeros(n_windows * L * F, np.uint16)
    cell = np.arange(n_windows * L, dtype=np.int64).reshape(n_windows, L) * F
    for _ in range(coverage):
        strand = rng.integers(0, 2, size=(n_windows, L))
        b = base.copy()
        wrong = rng.random((n_windows, L)) < base_error
        b[wrong] = rng.integers(0, 5, size=int(wrong.sum()))
        r = rl.copy()
        off = rng.random((n_windows, L)) < rl_error
        step = np.where(rng.random(int(off.sum())) < np.where(rl[off] >= 5, 0.75, 0.5), -1, 1)
        r[off] = np.clip(r[off] + step, 1, 10)
        r[b == 0] = 0
        r[(b > 0) & (r == 0)] = 1
        feat = np.where(b == 0, strand * 45 + 44, strand * 45 + (b - 1) * 11 + r)
        votes[(cell + feat).ravel()] += 8          # one vote per (window, position) and read: the indices are distinct
    img = np.minimum(votes, 255).astype(np.uint8).reshape(n_windows, L, F)
    return img, base.astype(np.uint8), rl.astype(np.uint8)
