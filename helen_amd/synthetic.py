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
    img = _read_votes(base.ravel(), rl.ravel(), rng, coverage, base_error, rl_error).reshape(n_windows, L, F)
    return img, base.astype(np.uint8), rl.astype(np.uint8)


def _read_votes(base, rl, rng, coverage=14, base_error=0.12, rl_error=0.35):
    """The noise process of make_pileup_task on flat arrays of true labels: -> uint8 [len(base), 90] feature rows."""
    F = ImageSizeOptions.IMAGE_HEIGHT
    n = base.shape[0]
    votes = np.zeros(n * F, np.uint16)
    cell = np.arange(n, dtype=np.int64) * F
    for _ in range(coverage):
        strand = rng.integers(0, 2, size=n)
        b = base.copy()
        wrong = rng.random(n) < base_error
        b[wrong] = rng.integers(0, 5, size=int(wrong.sum()))
        r = rl.copy()
        off = rng.random(n) < rl_error
        step = np.where(rng.random(int(off.sum())) < np.where(rl[off] >= 5, 0.75, 0.5), -1, 1)
        r[off] = np.clip(r[off] + step, 1, 10)
        r[b == 0] = 0
        r[(b > 0) & (r == 0)] = 1
        feat = np.where(b == 0, strand * 45 + 44, strand * 45 + (b - 1) * 11 + r)
        votes[cell + feat] += 8          # one vote per position and read: the indices are distinct
    return np.minimum(votes, 255).astype(np.uint8).reshape(n, F)


# ---- a simulated assembly: what `helen polish` is checked on end to end (BASELINE.json configs[4] stand-in) ----
_NOISE_VARIANTS = 64          # distinct noisy feature rows per true (base, run length) class


class SimContig(object):
    """One contig of a simulated draft assembly in MarginPolish's row space.  A ROW is a position key (pos, indx, split)
    of dataloader_predict.py:69 / Stitch.py:221-231: (p, 0, 0) = draft position p, (p, 0, 1) = the continuation of a run
    longer than 10, (p, 1, 0) = an insert column after p.  Every row has a true base label (0 = nothing there, 1..4 =
    A, C, G, T) and a true run length; the polished truth is the rows decoded in key order, base x run length.

    The contig is cut into REGIONS of `region_positions` draft positions, consecutive regions sharing `overlap` of them
    (Options.py:17 SEQ_OVERLAP); a region's rows are cut into images of at most 1000 rows (`feature_chunk_idx` 0, 1, ...),
    consecutive images sharing `chunk_overlap_rows` rows, the last one short.  `holes` = region indices left out (a
    stretch no image covers: stitch fills it with N x 10, Stitch.py:176-187).

    contig_start / contig_end of a region are offsets in the EXPANDED draft (run lengths spelled out), not row positions:
    stitch takes `running_end - this_start` as the number of BASES two neighbours share (Stitch.py:141-147), so the
    interval has to be in the sequence's own coordinates for the overlap alignment to see the shared stretch."""

    def __init__(self, name, n_positions, seed, region_positions=2400, overlap=200, insert_rate=0.10,
                 chunk_overlap_rows=16, holes=()):
        rng = np.random.default_rng(seed)
        n = int(n_positions)
        L = ImageSizeOptions.SEQ_LENGTH
        base0 = rng.integers(1, 5, size=n)
        base0[rng.random(n) < 0.03] = 0                       # the draft has a base the truth has not
        rl0 = np.minimum(10, rng.geometric(0.55, size=n))
        rl0[base0 == 0] = 0
        has_split = (rl0 == 10) & (rng.random(n) < 0.5)
        has_insert = rng.random(n) < insert_rate
        per = 1 + has_split.astype(np.int64) + has_insert.astype(np.int64)
        first = np.concatenate([[0], np.cumsum(per)])         # row index of (p, 0, 0); first[n] = number of rows
        rows = int(first[n])
        pos = np.repeat(np.arange(n, dtype=np.int64), per)
        indx = np.zeros(rows, np.int64)
        split = np.zeros(rows, np.int64)
        base = np.zeros(rows, np.int64)
        rl = np.zeros(rows, np.int64)
        base[first[:n]] = base0
        rl[first[:n]] = rl0
        at = first[:n][has_split] + 1
        split[at] = 1
        base[at] = base0[has_split]
        rl[at] = np.minimum(10, rng.geometric(0.55, size=at.shape[0]))
        at = (first[:n] + has_split)[has_insert] + 1
        indx[at] = 1
        ins_base = rng.integers(1, 5, size=at.shape[0])
        ins_base[rng.random(at.shape[0]) < 0.7] = 0
        base[at] = ins_base
        ins_rl = np.minimum(10, rng.geometric(0.55, size=at.shape[0]))
        ins_rl[ins_base == 0] = 0
        rl[at] = ins_rl
        self.name, self.n_positions = name, n
        self.position = np.stack([pos, indx, split], axis=1)                       # int64 [rows, 3], in key order
        self.label_base, self.label_rle = base.astype(np.uint8), rl.astype(np.uint8)
        # which of the _NOISE_VARIANTS noisy renderings of its class a row shows
        self.code = ((base * 11 + rl) * _NOISE_VARIANTS + rng.integers(0, _NOISE_VARIANTS, size=rows)).astype(np.int32)
        step = region_positions - overlap
        expanded = np.concatenate([[0], np.cumsum(np.where(base0 > 0, rl0, 1))])     # draft offset of position p
        self.regions = []            # (contig_start, contig_end, first row, end row)
        self.windows = []            # (region index, feature_chunk_idx, first row, end row)
        k = 0
        while True:
            start = k * step
            if k > 0 and start >= n - overlap:
                break
            end = min(start + region_positions, n)
            if k not in holes:
                lo, hi = int(first[start]), int(first[end])
                self.regions.append((int(expanded[start]), int(expanded[end]), lo, hi))
                c, at = 0, lo
                while True:
                    self.windows.append((len(self.regions) - 1, c, at, min(at + L, hi)))
                    if at + L >= hi:
                        break
                    at += L - chunk_overlap_rows
                    c += 1
            k += 1

    def truth(self):
        """The polished sequence a perfect caller would produce for the whole contig."""
        letters = np.frombuffer(b"NACGT", np.uint8)
        return np.repeat(letters[self.label_base], self.label_rle).tobytes().decode()


def _noise_bank(seed=77):
    """uint8 [(5 * 11) * _NOISE_VARIANTS + 1, 90]: row (base * 11 + rl) * _NOISE_VARIANTS + v = the v-th noisy rendering of
    a position whose truth is (base, rl), through the read-vote process of make_pileup_task; the last row is all zero
    (what the reader pads short images with)."""
    cls = np.arange(5 * 11)
    base = np.repeat(cls // 11, _NOISE_VARIANTS)
    rl = np.repeat(cls % 11, _NOISE_VARIANTS)
    valid = ((base == 0) & (rl == 0)) | ((base > 0) & (rl > 0))
    bank = np.zeros((base.shape[0] + 1, ImageSizeOptions.IMAGE_HEIGHT), np.uint8)
    rng = np.random.default_rng(seed)
    bank[:-1][valid] = _read_votes(base[valid], rl[valid], rng)
    return bank


def assembly_contigs(spec, seed=20260929):
    """spec = [(name, n_positions, {SimContig keyword: value}), ...] -> [SimContig]; contig k is seeded seed + k."""
    return [SimContig(name, n, seed + k, **kw) for k, (name, n, kw) in enumerate(spec)]


def contig_windows(contig, bank, lo=0, hi=None):
    """Images lo..hi of one SimContig as the arrays a file writer takes:
    -> (starts, ends, chunk ids int64 [n], lengths int32 [n], images uint8 [n, 1000, 90], positions int64 [n, 1000, 3]);
    rows past an image's length are zero / (-1, -1, -1), which no writer stores."""
    L = ImageSizeOptions.SEQ_LENGTH
    wins = contig.windows[lo:hi]
    n = len(wins)
    starts = np.array([contig.regions[w[0]][0] for w in wins], np.int64)
    ends = np.array([contig.regions[w[0]][1] for w in wins], np.int64)
    chunks = np.array([w[1] for w in wins], np.int64)
    first = np.array([w[2] for w in wins], np.int64)
    lengths = np.array([w[3] - w[2] for w in wins], np.int32)
    row = first[:, None] + np.arange(L, dtype=np.int64)[None, :]
    live = np.arange(L)[None, :] < lengths[:, None]
    row = np.where(live, row, 0)
    codes = np.where(live, contig.code[row], bank.shape[0] - 1)
    images = bank[codes]
    positions = np.where(live[:, :, None], contig.position[row], -1)
    return starts, ends, chunks, lengths, images, positions


def _write_assembly_files(args):
    directory, spec, n_files, seed, blocks, direct, only, gzip = args
    made = write_assembly_dir(directory, spec, n_files, seed, blocks, direct, only_files=only, gzip=gzip)
    return made["files"], made["windows"], made["regions"], made["windows_per_file"], made["regions_per_file"]


def assembly_spec(windows, n_files, contigs_per_file=2, region_positions=2400, overlap=200, name="chr%02d_sim"):
    """A simulated assembly of about `windows` images for a benchmark: n_files * contigs_per_file contigs of equal length,
    contig k in file k % n_files (three images per 2400-position region, the last one short)."""
    n_contigs = n_files * contigs_per_file
    regions = max(1, windows // (3 * n_contigs))
    positions = regions * (region_positions - overlap) + overlap
    return [(name % k, positions, {"region_positions": region_positions, "overlap": overlap}) for k in range(n_contigs)]


def write_assembly_dir(directory, spec, n_files=1, seed=20260929, blocks=None, direct=False, only_files=None, processes=0, gzip=None):
    """A directory of MarginPolish-shaped image files for the simulated assembly `spec` (see assembly_contigs).  Each
    contig's regions are cut into blocks[k] (default 1) runs of consecutive regions; the blocks, in contig order, go to
    the files round-robin -- all images of a region share a file, a contig may span several.  Images are named
    <contig>-<contig_start>-<contig_end>-<feature_chunk_idx>.  direct=True writes through the emitter of libhelen_io.so
    (the benchmark's 300 k-window inputs in seconds) instead of libhdf5; only_files = the file indices this caller writes
    (several ranks of a benchmark each write their share); processes = N > 1 writes the files in N worker processes;
    gzip = 1..9 (with direct=False) stores image and position chunked and deflated, as an h5py writer with compression="gzip".
    -> {"files": [paths], "windows": total images, "regions": total, "truth": {contig: sequence} (None when only_files is given),
        "windows_per_file": [...]}"""
    os.makedirs(directory, exist_ok=True)
    blocks = list(blocks) if blocks is not None else [1] * len(spec)
    if processes and processes > 1:
        import concurrent.futures
        import multiprocessing as mp
        todo = sorted(range(n_files) if only_files is None else only_files)
        with concurrent.futures.ProcessPoolExecutor(min(processes, len(todo)), mp_context=mp.get_context("spawn")) as ex:
            parts = list(ex.map(_write_assembly_files, [(directory, spec, n_files, seed, blocks, direct, [fi], gzip) for fi in todo]))
        counts, region_counts = [0] * n_files, [0] * n_files
        for fi, p in zip(todo, parts):
            counts[fi] = p[3][fi]
            region_counts[fi] = p[4][fi]
        # (a worker renders every contig that touches its file: its own totals count those contigs whole)
        return {"files": [f for p in parts for f in p[0]], "windows": sum(counts), "regions": sum(region_counts), "truth": None,
                "windows_per_file": counts, "regions_per_file": region_counts}
    bank = _noise_bank()
    # which file a block goes to depends only on the blocks before it: block j -> file j % n_files
    paths = [os.path.join(directory, "assembly_images_%04d.h5" % fi) for fi in range(n_files)]
    mine = set(range(n_files)) if only_files is None else set(only_files)
    per_file = [[] for _ in range(n_files)]     # (contig, first image, end image)
    regions_per_file = [0] * n_files
    truth, j, windows, regions = {}, 0, 0, 0
    for k, (name, n, kw) in enumerate(spec):
        targets = [(j + b) % n_files for b in range(blocks[k])]
        j += blocks[k]
        if only_files is not None and not (set(targets) & mine):
            continue
        contig = SimContig(name, n, seed + k, **kw)
        if only_files is None:
            truth[name] = contig.truth()
        windows += len(contig.windows)
        regions += len(contig.regions)
        # cut at region boundaries, about equal numbers of regions per block
        region_of = np.array([w[0] for w in contig.windows])
        for b in range(blocks[k]):
            r_lo = len(contig.regions) * b // blocks[k]
            r_hi = len(contig.regions) * (b + 1) // blocks[k]
            lo, hi = int(np.searchsorted(region_of, r_lo)), int(np.searchsorted(region_of, r_hi))
            if hi > lo and targets[b] in mine:
                per_file[targets[b]].append((contig, lo, hi))
                regions_per_file[targets[b]] += r_hi - r_lo
    counts = []
    for fi in range(n_files):
        counts.append(sum(hi - lo for _, lo, hi in per_file[fi]))
        if fi not in mine or not per_file[fi]:
            continue
        parts = [(c.name,) + contig_windows(c, bank, lo, hi) for c, lo, hi in per_file[fi]]
        names = [p[0] for p in parts for _ in range(p[1].shape[0])]
        cat = [np.concatenate([p[i] for p in parts]) for i in range(1, 7)]
        if direct:
            from . import native_io
            native_io.emit_image_windows(paths[fi], names, *cat)
        else:
            _write_windows(paths[fi], names, *cat, gzip=gzip)
    return {"files": [p for fi, p in enumerate(paths) if per_file[fi] and fi in mine], "windows": windows, "regions": regions,
            "truth": truth if only_files is None else None, "windows_per_file": counts, "regions_per_file": regions_per_file}


def _write_windows(path, names, starts, ends, chunks, lengths, images, positions, gzip=None):
    with hdf5.File(path, "w") as f:
        for i, contig in enumerate(names):
            L = int(lengths[i])
            base = "images/%s-%d-%d-%d/" % (contig, starts[i], ends[i], chunks[i])
            f.write(base + "contig", contig)
            f.write(base + "contig_start", np.array([starts[i]], np.int64))
            f.write(base + "contig_end", np.array([ends[i]], np.int64))
            f.write(base + "feature_chunk_idx", np.array([chunks[i]], np.int64))
            f.write(base + "image", images[i, :L], np.uint8, chunks=(min(256, L), images.shape[2]) if gzip else None, gzip=gzip)
            f.write(base + "position", positions[i, :L], np.int64, chunks=(L, 3) if gzip else None, gzip=gzip)


# the simulated assembly of the chained `polish` parity fixture (tests/golden/make_golden_polish.py): three contigs in
# three files -- 2400-position regions of three images with a hole; 1000-position regions of two; one region of thirteen
# images (chunk ids 10..12 sort before 2 as strings, Stitch.py:211); a contig shorter than one image
POLISH_CASE = [
    ("ctgA", 30000, {"holes": (5,)}),
    ("ctgB.long_region", 12000, {"region_positions": 11000}),
    ("scaffold_3|tiny", 700, {}),
    ("ctgC", 20000, {"region_positions": 1000, "chunk_overlap_rows": 0}),
]
POLISH_CASE_BLOCKS = [2, 1, 1, 2]
POLISH_CASE_FILES = 3
