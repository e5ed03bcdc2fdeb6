"""The direct scanner of MarginPolish image files (helen_amd/csrc/h5scan.h) against libhdf5: the same batch read
through the scanner and through the library (HELEN_IO_READER=libhdf5, in a child process) must be byte-identical
for plain files and for every storage variant; files the scanner does not take (chunked / filtered datasets) and
damaged files must end up with libhdf5's answer or libhdf5's error, never with a guess."""
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


def test_scanner_takes_the_variants_it_can_and_leaves_the_rest_to_libhdf5(tmp_path):
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
    assert list(fast["contig"][:2]) == ["chr0quoted", "chr1quoted"]      # quotes stripped as the reader does
    assert tuple(fast["counts"]) == (6, 6)          # contiguous file: scanner; chunked + deflated file: libhdf5


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
        native_io.close_readers() if hasattr(native_io, "close_readers") else None
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
