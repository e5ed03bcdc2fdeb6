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
                     chunks_per_region=1, labels=None, gzip=None):
    """Write `images` (uint8 [n, 1000, 90]) as n images of one file.  Window k covers
    contig_start = 800*k .. +1000 (SEQ_OVERLAP 200, Options.py:17); `lengths[i] < 1000` stores a
    short image (the reader pads it).  labels = (label_base, label_run_length) uint8 [n, 1000] makes
    it a labeled file as the evaluation loader reads it (models/dataloader.py:59-61); gzip = 1..9 stores
    image and position chunked + deflated (as an h5py writer with compression="gzip" would)."""
    n = images.shape[0]
    with hdf5.File(path, "w") as f:
        for i in range(n):
            k = first_window + i
            region = k // chunks_per_region
            chunk = k % chunks_per_region
            start = 800 * region
            L = int(lengths[i]) if lengths is not None else ImageSizeOptions.SEQ_LENGTH
            name = "%s-%d-%d-%d" % (contig, start, start + 1000, chunk)
            base = "images/" + name + "/"
            f.write(base + "contig", contig)
            f.write(base + "contig_start", np.array([start], np.int64))
            f.write(base + "contig_end", np.array([start + 1000], np.int64))
            f.write(base + "feature_chunk_idx", np.array([chunk], np.int64))
            f.write(base + "image", images[i, :L], np.uint8, chunks=(256, 90) if gzip else None, gzip=gzip)
            pos = np.zeros((L, 3), np.int64)
            pos[:, 0] = start + np.arange(L)
            f.write(base + "position", pos, np.int64, chunks=(L, 3) if gzip else None, gzip=gzip)
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
