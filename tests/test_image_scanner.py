"""The direct scanner of MarginPolish image files (helen_amd/csrc/h5scan.h) against libhdf5: the same batch read
through the scanner and through the library (HELEN_IO_READER=libhdf5, in a child process) must be byte-identical
for plain files and for every storage variant -- contiguous, chunked, deflated, shuffled, checksummed, old and new file
format; what the scanner still declines (a paged chunk index) and damaged files must end up with libhdf5's answer or
libhdf5's error, never with a guess."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helen_amd import hdf5, native_io
from helen_amd.weights import make_images

pytestmark = pytest.mark.skipif(not (hdf5.available() and native_io.available()),
                                reason="libhdf5 / libhelen_io.so not available")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from helen_amd import native_io
from helen_amd.sequence_dataset import SequenceDataset, _load_batch
ds = SequenceDataset(None, file_list=%(files)r)
b = _load_batch(ds.all_images)
np.savez(%(out)r, images=b.images, positions=b.positions, start=b.contig_start, end=b.contig_end,
         chunk=b.chunk_id, contig=np.array(b.contig), names=np.array([n for _, n in ds.all_images]),
         counts=np.array(native_io.reader_counts()))
'''


def _read(files, out, reader):
    env = dict(os.environ)
    if reader:
        env["HELEN_IO_READER"] = reader
    else:
        env.pop("HELEN_IO_READER", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "files": list(files), "out": out}], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(out))


def _same(a, b):
    for k in ("images", "positions", "start", "end", "chunk", "contig", "names"):
        assert np.array_equal(a[k], b[k]), k


def test_scanner_equals_libhdf5_on_plain_files(tmp_path):
    from helen_amd.synthetic import write_image_dir
    files = write_image_dir(str(tmp_path / "img"), 700, n_files=2, short_every=7)   # > 512 images: two B-tree levels
    fast = _read(files, str(tmp_path / "fast.npz"), None)
    lib = _read(files, str(tmp_path / "lib.npz"), "libhdf5")
    _same(fast, lib)
    assert tuple(fast["counts"]) == (700, 0) and tuple(lib["counts"]) == (0, 700)
    first = [n for n in fast["names"][:350]]                        # .keys() order within a file
    assert first == sorted(first)


def test_scanner_reads_mixed_types_contiguous_and_packed(tmp_path):
    img = make_images(6, seed=5)
    rng = np.random.default_rng(2)
    plain, packed = str(tmp_path / "plain.h5"), str(tmp_path / "packed.h5")
    for path, store in ((plain, {}), (packed, dict(gzip=4, shuffle=True))):
        with hdf5.File(path, "w") as f:
            for i in range(6):
                L = [1000, 613, 1, 1000, 999, 1000][i]
                base = "images/w%d/" % i
                f.write(base + "contig", "chr%d'quoted'" % i, string=["fixed", "vlen", "scalar", "vlen_scalar", "fixed", "vlen"][i])
                ints = [np.int64, np.int32, np.uint16, np.uint64, np.int16, np.uint32][i]
                for name, val in (("contig_start", 80 * i), ("contig_end", 80 * i + 1000), ("feature_chunk_idx", i % 3)):
                    f.write(base + name, np.array(val if i % 2 else [val], ints))
                it = [np.uint8, np.uint16, np.int32, np.float32, np.int64, np.float64][i]
                f.write(base + "image", img[i, :L].astype(it), it, chunks=(min(L, 128), 90) if store else None, **store)
                pt = [np.int64, np.int32, np.uint32, np.uint64, np.int16, np.int64][i]
                pos = np.stack([100 + np.arange(L), rng.integers(0, 3, L), rng.integers(0, 2, L)], 1)
                f.write(base + "position", pos.astype(pt), pt, chunks=(L, 3) if store else None, **store)
    fast = _read([plain, packed], str(tmp_path / "fast.npz"), None)
    lib = _read([plain, packed], str(tmp_path / "lib.npz"), "libhdf5")
    _same(fast, lib)
    # np.array2string(...).replace("'", '') of the reference's reader: a name with single quotes is printed in double
    # quotes, which stay (pinned against the reference's own reader in tests/test_host_io.py)
    assert list(fast["contig"][:2]) == ['"chr0quoted"', '"chr1quoted"']
    assert tuple(fast["counts"]) == (12, 0)         # both through the scanner (round 4: chunked + shuffled + deflated too)
    assert tuple(lib["counts"]) == (0, 12)


VARIANTS = {
    "chunked_rows": dict(chunks=(100, 90)),
    "chunked_one": dict(chunks=(1000, 90)),
    "chunked_ragged": dict(chunks=(384, 90)),
    "gzip1": dict(gzip=1),
    "gzip9": dict(gzip=9),
    "shuffle_gzip": dict(gzip=4, shuffle=True),
    "shuffle_only": dict(shuffle=True),
    "fletcher32": dict(fletcher32=True),
    "gzip_fletcher32": dict(gzip=4, shuffle=True, fletcher32=True),
    "vlen_names": dict(string="vlen"),
    "latest": dict(libver="latest"),
    "latest_vlen": dict(libver="latest", string="vlen"),
    "latest_chunked": dict(libver="latest", chunks=(100, 90)),          # fixed-array chunk index
    "latest_single_chunk": dict(libver="latest", chunks=(1000, 90)),    # single-chunk index (short images: one chunk too)
    "latest_gzip": dict(libver="latest", gzip=4, shuffle=True),         # filtered fixed array / filtered single chunk
    "latest_gzip_fletcher32": dict(libver="latest", gzip=6, fletcher32=True),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_scanner_equals_libhdf5_on_storage_variants(tmp_path, variant):
    """Every way of storing an image file that libhdf5 / h5py offer short of unlimited dimensions: the scanner reads all of
    them itself (no image goes through libhdf5) and returns what libhdf5 returns.  40 images, so that a new-style
    `images` group is DENSE (fractal heap + version 2 B-tree) while each image's own group keeps its links in its header;
    short images (613, 1 row) make ragged and single chunks."""
    from helen_amd.synthetic import write_image_file
    n = 40
    img = make_images(n, seed=9, mode="pileup")
    lengths = np.full(n, 1000)
    lengths[5::7] = 613
    lengths[3] = 1
    path = str(tmp_path / (variant + ".h5"))
    write_image_file(path, img, lengths=lengths, **VARIANTS[variant])
    fast = _read([path], str(tmp_path / "fast.npz"), None)
    lib = _read([path], str(tmp_path / "lib.npz"), "libhdf5")
    _same(fast, lib)
    assert tuple(fast["counts"]) == (n, 0) and tuple(lib["counts"]) == (0, n)
    assert list(fast["names"]) == sorted(fast["names"])
    # the direct scanner ALONE (the fallback switched off): listing, per-name reads and positional reads
    alone = _read([path], str(tmp_path / "alone.npz"), "direct")
    _same(alone, lib)


def test_a_large_new_style_group(tmp_path):
    """3,000 images in a libver=latest file: a fractal heap with indirect blocks and a version 2 B-tree with internal
    nodes.  Names in libhdf5's order, every image read."""
    from helen_amd.synthetic import write_image_file
    n = 3000
    img = make_images(n, seed=4, mode="pileup")
    path = str(tmp_path / "many.h5")
    write_image_file(path, img, lengths=np.full(n, 2), libver="latest")
    fast = _read([path], str(tmp_path / "fast.npz"), "direct")
    lib = _read([path], str(tmp_path / "lib.npz"), "libhdf5")
    _same(fast, lib)
    assert tuple(fast["counts"]) == (n, 0)


def test_what_the_scanner_declines_goes_to_libhdf5(tmp_path):
    """A chunk index the scanner does not walk (libver=latest, 2,000 chunks per image: a PAGED fixed array): those images
    are read by libhdf5, the others of the same run by the scanner, the batch is libhdf5's."""
    from helen_amd.synthetic import write_image_file
    img = make_images(4, seed=2, mode="pileup")
    paged, plain = str(tmp_path / "a_paged.h5"), str(tmp_path / "b_plain.h5")
    write_image_file(paged, img[:2], libver="latest", chunks=(1, 45))
    write_image_file(plain, img[2:], first_window=2)
    fast = _read([paged, plain], str(tmp_path / "fast.npz"), None)
    lib = _read([paged, plain], str(tmp_path / "lib.npz"), "libhdf5")
    _same(fast, lib)
    assert tuple(fast["counts"]) == (2, 2)
    assert native_io.index_images(paged) == (2, False)       # the file is the scanner's (its groups are); its datasets are not
    n = 4
    images = np.zeros((n, 1000, 90), np.uint8)
    positions = np.zeros((n, 1000, 3), np.int64)
    meta = np.zeros((n, 3), np.int64)
    contigs = np.zeros((n, native_io.NAME_BYTES), np.uint8)
    assert native_io.read_image_runs([(paged, 0, 2), (plain, 0, 2)], 3, images, positions, meta, contigs) == 2
    assert np.array_equal(images, lib["images"]) and np.array_equal(positions, lib["positions"])


def test_damaged_files_are_libhdf5s_business(tmp_path):
    from helen_amd.sequence_dataset import SequenceDataset, _load_batch
    from helen_amd.synthetic import write_image_dir
    files = write_image_dir(str(tmp_path / "img"), 24, n_files=1)
    raw = open(files[0], "rb").read()
    # truncated in the middle of the data, garbage in the middle, not HDF5 at all: whatever happens, it is an
    # error from the readers (or, for damage that misses every structure that is read, a normal read) -- never a
    # crash and never silently different data
    ok = _load_batch(SequenceDataset(None, file_list=files).all_images)
    for k, blob in enumerate((raw[:len(raw) // 2], raw[:2000] + os.urandom(4000) + raw[6000:], b"not an hdf5 file" * 100)):
        bad = str(tmp_path / ("bad%d.h5" % k))
        open(bad, "wb").write(blob)
        native_io.close_readers()
        try:
            ds = SequenceDataset(None, file_list=[bad])
            got = _load_batch(ds.all_images)
        except (IOError, OSError, ValueError, RuntimeError):
            continue
        for i, (_, name) in enumerate(ds.all_images):
            j = [n for _, n in SequenceDataset(None, file_list=files).all_images].index(name)
            assert np.array_equal(got.images[i], ok.images[j])


def test_directly_emitted_image_files_equal_libhdf5_written_ones(tmp_path):
    """helen_amd.synthetic.write_image_dir(direct=True) (benchmark inputs, through h5emit.h) stores what the libhdf5
    path stores: h5diff finds no difference, and both readers return the same batch from either."""
    import shutil
    from helen_amd.synthetic import write_image_dir
    a = write_image_dir(str(tmp_path / "lib"), 300, n_files=2, short_every=7)
    b = write_image_dir(str(tmp_path / "direct"), 300, n_files=2, short_every=7, direct=True)
    h5diff = shutil.which("h5diff") or "/opt/conda/bin/h5diff"
    if os.path.exists(h5diff):
        for x, y in zip(a, b):
            r = subprocess.run([h5diff, x, y], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
    for reader in (None, "libhdf5"):
        _same(_read(a, str(tmp_path / "a.npz"), reader), _read(b, str(tmp_path / "b.npz"), reader))


FUZZ_CHILD = r'''
import hashlib, os, sys, numpy as np
sys.path.insert(0, %(root)r)
from helen_amd import native_io
paths = %(paths)r
for k in range(%(first)d, len(paths)):
    try:
        names = native_io.list_images(paths[k])
        if names is None:
            print(k, "noimages", flush=True)
            continue
        n = len(names)
        images = np.zeros((n, 1000, 90), np.uint8); positions = np.zeros((n, 1000, 3), np.int64)
        meta = np.zeros((n, 3), np.int64); contigs = np.zeros((n, native_io.NAME_BYTES), np.uint8)
        if n:
            native_io.read_images(paths[k], names, images, positions, meta, contigs)
        h = hashlib.sha1()
        for a in (images, positions, meta, contigs):
            h.update(a.tobytes())
        h.update("\n".join(names).encode())
        print(k, "ok", h.hexdigest(), flush=True)
    except (IOError, OSError, ValueError) as e:
        print(k, "error", flush=True)
'''


def _fuzz_run(paths, reader, tolerate_crashes):
    """{index: (status, digest)} of reading every file with one reader mode; a crashing reader is restarted behind
    the file that killed it when `tolerate_crashes` (libhdf5 1.10 on damaged files), else it fails the test."""
    env = dict(os.environ, HELEN_IO_READER=reader)
    out, first = {}, 0
    while first < len(paths):
        r = subprocess.run([sys.executable, "-c", FUZZ_CHILD % {"root": ROOT, "paths": list(paths), "first": first}],
                           env=env, capture_output=True, text=True, timeout=900)
        for line in r.stdout.splitlines():
            parts = line.split()
            out[int(parts[0])] = (parts[1], parts[2] if len(parts) > 2 else "")
        if r.returncode == 0:
            break
        assert tolerate_crashes, "reader '%s' died (rc %d) on %s\n%s" % (
            reader, r.returncode, paths[max(out) + 1 if out else first], r.stderr[-1500:])
        dead = max(out) + 1 if out else first
        out[dead] = ("crash", "")
        first = dead + 1
    return out


def test_scanner_survives_damaged_metadata(tmp_path):
    """~900 mutants of two small image files (one written by libhdf5, one by the emitter): bytes flipped, 64-bit
    fields replaced by 0 / ~0 / huge / random values, 16-bit counts maxed out, files truncated.  The scanner alone
    (HELEN_IO_READER=direct: no fallback) must never crash or hang; and whenever it returns a batch for a file
    libhdf5 also reads, the two batches are byte-identical."""
    rng = np.random.default_rng(77)
    img = make_images(8, seed=3)
    bases = []
    for kind in ("lib", "direct"):
        path = str(tmp_path / (kind + ".h5"))
        lengths = [5, 12, 3, 9, 1, 7, 12, 4]
        if kind == "lib":
            with hdf5.File(path, "w") as f:
                for i, L in enumerate(lengths):
                    b = "images/img_%02d/" % i
                    f.write(b + "contig", "chrF", string="vlen" if i % 2 else "fixed")
                    f.write(b + "contig_start", np.array([100 * i], np.int64))
                    f.write(b + "contig_end", np.array([100 * i + L], np.int32))
                    f.write(b + "feature_chunk_idx", np.array([i % 2], np.uint8))
                    f.write(b + "image", img[i, :L], np.uint8)
                    f.write(b + "position", np.stack([np.arange(L), np.zeros(L), np.zeros(L)], 1).astype(np.int64), np.int64)
        else:
            native_io.emit_images(path, "chrF", np.arange(8) * 100, np.arange(8) % 2, np.array(lengths, np.int32),
                                  img)
        bases.append(open(path, "rb").read())
    paths = []
    for b, raw in enumerate(bases):
        size = len(raw)
        for k in range(450):
            m = bytearray(raw)
            how = k % 5
            if how == 0:                                   # a few flipped bytes
                for _ in range(int(rng.integers(1, 5))):
                    m[int(rng.integers(0, size))] ^= int(rng.integers(1, 256))
            elif how == 1:                                 # an aligned 64-bit field: 0, ~0, just past the end, random
                o = int(rng.integers(0, size // 8)) * 8
                v = [0, 2 ** 64 - 1, size + int(rng.integers(0, 64)), int(rng.integers(0, 2 ** 63))][int(rng.integers(0, 4))]
                m[o:o + 8] = int(v).to_bytes(8, "little")
            elif how == 2:                                 # a 16-bit field maxed out / zeroed (entry and message counts)
                o = int(rng.integers(0, size // 2)) * 2
                m[o:o + 2] = b"\xff\xff" if rng.integers(0, 2) else b"\0\0"
            elif how == 3:                                 # truncated
                m = m[:int(rng.integers(100, size))]
            else:                                          # a pointer redirected to another structure of the file
                o = int(rng.integers(0, size // 8)) * 8
                m[o:o + 8] = (int(rng.integers(0, size // 8)) * 8).to_bytes(8, "little")
            p = str(tmp_path / ("m%d_%03d.h5" % (b, k)))
            open(p, "wb").write(bytes(m))
            paths.append(p)
    direct = _fuzz_run(paths, "direct", tolerate_crashes=False)
    assert len(direct) == len(paths)
    lib = _fuzz_run(paths, "libhdf5", tolerate_crashes=True)
    both = differ = 0
    lenient = []
    for k in range(len(paths)):
        if direct[k][0] == "ok" and lib.get(k, ("crash", ""))[0] == "ok":
            both += 1
            if direct[k][1] != lib[k][1]:
                differ += 1
                print("DIFFERENT:", paths[k])
        elif direct[k][0] == "ok":
            lenient.append(os.path.basename(paths[k]))
    stats = {s: sum(1 for v in direct.values() if v[0] == s) for s in ("ok", "error", "noimages")}
    print("scanner:", stats, " libhdf5 crashes:", sum(1 for v in lib.values() if v[0] == "crash"), " both ok:", both,
          " scanner only:", lenient)
    assert differ == 0
    assert both > 100 and stats["error"] > 100      # the mutants did hit structures that matter, and data that do not


FUZZ_PRED_CHILD = r'''
import hashlib, os, sys
sys.path.insert(0, %(root)r)
from helen_amd import native_io
paths = %(paths)r
for k in range(%(first)d, len(paths)):
    try:
        h = hashlib.sha1()
        for contig in ("ctgA", "ctgB"):
            listed = native_io.list_regions(paths[k], contig)
            h.update(repr(listed).encode())
            for name, st, en in (listed or []):
                h.update(native_io.region_sequence(paths[k], contig, name).encode())
        print(k, "ok", h.hexdigest(), flush=True)
    except (IOError, OSError, ValueError, UnicodeDecodeError) as e:
        print(k, "error", flush=True)
'''


def test_scanner_survives_damaged_prediction_files(tmp_path, monkeypatch):
    """The same treatment for the stitch side (helen_io_list_regions / helen_io_region_sequence): ~600 mutants of a
    small prediction file from either writer; the scanner alone never crashes, and agrees with libhdf5 wherever
    both return something."""
    global FUZZ_CHILD
    rng = np.random.default_rng(78)
    bases = []
    for kind in (None, "libhdf5"):
        if kind:
            monkeypatch.setenv("HELEN_IO_WRITER", kind)
        else:
            monkeypatch.delenv("HELEN_IO_WRITER", raising=False)
        path = str(tmp_path / ("pred_%s.hdf" % (kind or "direct")))
        w = native_io.Writer(path)
        n = 9
        names = ["ctgA" if i % 3 else "ctgB" for i in range(n)]
        meta = np.stack([np.arange(n) // 2 * 800, np.arange(n) // 2 * 800 + 1000, np.arange(n) % 2], 1).astype(np.int64)
        pos = np.zeros((n, 1000, 3), np.int64)
        pos[:, :, 0] = np.arange(1000)[None, :]
        w.write(native_io.pack_contigs(names), meta, pos, rng.integers(0, 5, (n, 1000), dtype=np.uint8),
                rng.integers(0, 3, (n, 1000), dtype=np.uint8))
        w.close()
        bases.append(open(path, "rb").read())
    paths = []
    for b, raw in enumerate(bases):
        size = len(raw)
        # metadata only: object headers and group structures sit between / behind the 14 KB data blocks
        meta_zones = [i for i in range(0, size - 8, 8) if raw[i:i + 4] in (b"TREE", b"SNOD", b"HEAP") or raw[i] == 1]
        for k in range(300):
            m = bytearray(raw)
            how = k % 4
            o = int(meta_zones[int(rng.integers(0, len(meta_zones)))]) + 8 * int(rng.integers(0, 12))
            o = min(o, size - 8)
            if how == 0:
                for _ in range(int(rng.integers(1, 4))):
                    m[min(size - 1, o + int(rng.integers(0, 64)))] ^= int(rng.integers(1, 256))
            elif how == 1:
                v = [0, 2 ** 64 - 1, size + int(rng.integers(0, 64)), int(rng.integers(0, 2 ** 63))][int(rng.integers(0, 4))]
                m[o:o + 8] = int(v).to_bytes(8, "little")
            elif how == 2:
                m[o:o + 2] = b"\xff\xff" if rng.integers(0, 2) else b"\0\0"
            else:
                m[o:o + 8] = (int(rng.integers(0, size // 8)) * 8).to_bytes(8, "little")
            p = str(tmp_path / ("q%d_%03d.hdf" % (b, k)))
            open(p, "wb").write(bytes(m))
            paths.append(p)
    pristine = []
    for b, raw in enumerate(bases):
        p = str(tmp_path / ("q%d_pristine.hdf" % b))
        open(p, "wb").write(raw)
        pristine.append(p)
    saved = FUZZ_CHILD
    try:
        FUZZ_CHILD = FUZZ_PRED_CHILD
        truth = _fuzz_run(pristine, "direct", tolerate_crashes=False)
        direct = _fuzz_run(paths, "direct", tolerate_crashes=False)
        lib = _fuzz_run(paths, "libhdf5", tolerate_crashes=True)
    finally:
        FUZZ_CHILD = saved
    assert len(direct) == len(paths) and truth[0][0] == truth[1][0] == "ok"
    both = differ = 0
    for k in range(len(paths)):
        if direct[k][0] == "ok" and lib.get(k, ("crash", ""))[0] == "ok":
            both += 1
            # (a damaged cache field of a symbol table entry makes libhdf5 report a contig as absent; the scanner
            # does not read that field and returns what the undamaged file holds: that is not a disagreement)
            if direct[k][1] != lib[k][1] and direct[k][1] != truth[k // 300][1]:
                differ += 1
                print("DIFFERENT:", paths[k])
    stats = {s: sum(1 for v in direct.values() if v[0] == s) for s in ("ok", "error")}
    print("scanner:", stats, " libhdf5 crashes:", sum(1 for v in lib.values() if v[0] == "crash"), " both ok:", both)
    assert differ == 0
    assert both > 50 and stats["error"] > 50
