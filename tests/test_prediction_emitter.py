"""The direct HDF5 emitter of the prediction writer (helen_amd/csrc/h5emit.h) against libhdf5 itself: files it
writes are read back through libhdf5 (ctypes binding and, when present, the h5diff / h5ls tools), compared with
what the libhdf5-based writer produces from the same calls, and a 70,000-member group (three B-tree levels) is
walked member by member."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helen_amd import hdf5, native_io

pytestmark = pytest.mark.skipif(not (hdf5.available() and native_io.available()),
                                reason="libhdf5 / libhelen_io.so not available")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5DIFF = shutil.which("h5diff") or "/opt/conda/bin/h5diff"
H5LS = shutil.which("h5ls") or "/opt/conda/bin/h5ls"


def _batch(n, rng, contigs=("ctgA", "c" * 255, "x")):
    names = [contigs[i % len(contigs)] for i in range(n)]
    meta = np.zeros((n, 3), np.int64)
    meta[:, 0] = (np.arange(n) // 13) * 800           # 13 chunk ids per region: '10', '11', '12' sort before '2'
    meta[:, 1] = meta[:, 0] + 1000
    meta[:, 2] = np.arange(n) % 13
    pos = rng.integers(-1, 2 ** 31, (n, 1000, 3)).astype(np.int64)
    return names, meta, pos, rng.integers(0, 5, (n, 1000), dtype=np.uint8), rng.integers(0, 11, (n, 1000), dtype=np.uint8)


def _write(path, batches, monkeypatch, which):
    if which:
        monkeypatch.setenv("HELEN_IO_WRITER", which)
    else:
        monkeypatch.delenv("HELEN_IO_WRITER", raising=False)
    w = native_io.Writer(path)
    for names, meta, pos, b, r in batches:
        w.write(native_io.pack_contigs(names), meta, pos, b, r)
    w.close()


def _walk(f, group="/", out=None):
    out = {} if out is None else out
    for k in f.keys(group):
        p = (group.rstrip("/") + "/" + k)
        try:
            sub = f.keys(p)
        except hdf5.Hdf5Error:
            sub = None
        if sub is None or p.rsplit("/", 1)[1] in ("position", "bases", "rles", "contig_start", "contig_end"):
            out[p] = f.read(p)
        else:
            _walk(f, p, out)
    return out


def test_emitted_file_equals_the_libhdf5_writers(tmp_path, monkeypatch):
    rng = np.random.default_rng(7)
    batches = [_batch(200, rng), _batch(57, rng)]
    batches.append(batches[0])                                    # repeats are skipped (DataStore.py:102-124)
    a, b = str(tmp_path / "emit.hdf"), str(tmp_path / "lib.hdf")
    _write(a, batches, monkeypatch, None)
    _write(b, batches, monkeypatch, "libhdf5")
    with hdf5.File(a) as fa, hdf5.File(b) as fb:
        wa, wb = _walk(fa), _walk(fb)
        assert list(wa) == list(wb) and len(wa) > 600            # same members in the same (name) order
        for k in wa:
            assert wa[k].dtype == wb[k].dtype and wa[k].shape == wb[k].shape and np.array_equal(wa[k], wb[k]), k
        i = fa.info("predictions/ctgA/ctgA-0-1000/12/position")
        assert i["layout"] == "contiguous" and i["size"] == 4 and not i["signed"] and i["shape"] == (1000, 3)
        # contigs take every third window: chunk ids 0, 3, 6, 9, 12 of the first region, in STRING order
        assert fa.keys("predictions/ctgA/ctgA-0-1000") == ["0", "12", "3", "6", "9", "contig_end", "contig_start"]
    if os.path.exists(H5DIFF):
        r = subprocess.run([H5DIFF, a, b], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr          # libhdf5's own tool: no differences


def test_an_unused_writer_leaves_a_file_without_predictions(tmp_path, monkeypatch):
    p = str(tmp_path / "none.hdf")
    _write(p, [], monkeypatch, None)
    with hdf5.File(p) as f:
        assert f.keys("/") == [] and "predictions" not in f


def test_stitch_reads_emitted_and_library_files_alike(tmp_path, monkeypatch, capfd):
    from helen_amd.stitch import perform_stitch
    rng = np.random.default_rng(3)
    names, meta, pos, b, r = _batch(90, rng, contigs=("chrS",))
    pos[:, :, 0] = np.arange(1000)[None, :] + meta[:, :1]
    pos[:, :, 1:] = 0
    fasta = []
    for which in (None, "libhdf5"):
        d = tmp_path / (which or "emit")
        d.mkdir()
        _write(str(d / "p_0.hdf"), [(names, meta, pos, b, r)], monkeypatch, which)
        out = perform_stitch(str(d), str(tmp_path / ("fa_" + (which or "emit"))), "asm", 2)
        fasta.append(open(out).read())
    capfd.readouterr()
    assert fasta[0] == fasta[1] and len(fasta[0]) > 1000


def test_large_groups_and_every_name_length(tmp_path):
    """70,000 members = 8,750 symbol table nodes under three B-tree levels; names of every length modulo 8; an empty
    group; all read back through libhdf5 member by member."""
    exe = str(tmp_path / "emit_many")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "helen_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "emit_many.cpp")])
    n = 70000
    path = str(tmp_path / "many.hdf")
    subprocess.check_call([exe, path, str(n)])
    want = sorted(str(i) + chr(ord("a") + i % 26) * (i % 9) for i in range(n))
    with hdf5.File(path) as f:
        assert f.keys("/") == ["mid"] and f.keys("mid") == ["answer", "empty", "many"]
        assert f.keys("mid/empty") == [] and int(f.read("mid/answer")) == 42
        assert f.keys("mid/many") == want                        # complete, and in strcmp order
        rng = np.random.default_rng(1)
        for i in list(rng.integers(0, n, 400)) + [0, 1, n - 1, n - 2]:
            name = str(i) + chr(ord("a") + i % 26) * (i % 9)
            assert int(f.read("mid/many/" + name)) == 3 * int(i) - 7, name      # by-name lookup through the B-tree
        assert ("mid/many/" + "nope") not in f and ("mid/many/" + want[-1] + "z") not in f
    if os.path.exists(H5LS):
        r = subprocess.run([H5LS, path + "/mid/many"], capture_output=True, text=True)
        assert r.returncode == 0 and len(r.stdout.splitlines()) == n


@pytest.mark.skipif(not os.path.exists("/dev/full"), reason="no /dev/full")
def test_a_full_disk_is_an_error_not_a_short_file(monkeypatch):
    """The emitter buffers 8 MiB and writes the group structures at close: a write that fails must surface as an
    IOError from write() or close(), never pass silently."""
    monkeypatch.delenv("HELEN_IO_WRITER", raising=False)
    rng = np.random.default_rng(0)
    names, meta, pos, b, r = _batch(8, rng)
    with pytest.raises(IOError):
        w = native_io.Writer("/dev/full")
        w.write(native_io.pack_contigs(names), meta, pos, b, r)
        w.close()


def test_contig_names_that_are_hdf5_paths(tmp_path, monkeypatch):
    """`predictions/{contig}/{contig}-{start}-{end}/...` is a PATH for the reference's h5py writer and for libhdf5:
    a contig named 'a/b' makes nested groups, empty components and '.' vanish.  The emitter builds the same tree
    (it used to store 'a/b' as ONE link name, which libhdf5 / h5py cannot open)."""
    rng = np.random.default_rng(11)
    batches = [_batch(40, rng, contigs=("plain", "a/b", "a/c//d", "./lead", "trail/"))]
    a, b = str(tmp_path / "emit.hdf"), str(tmp_path / "lib.hdf")
    _write(a, batches, monkeypatch, None)
    _write(b, batches, monkeypatch, "libhdf5")
    with hdf5.File(a) as fa, hdf5.File(b) as fb:
        wa, wb = _walk(fa), _walk(fb)
        assert list(wa) == list(wb) and len(wa) > 100
        for k in wa:
            assert wa[k].dtype == wb[k].dtype and np.array_equal(wa[k], wb[k]), k
        assert fa.keys("predictions") == fb.keys("predictions") == ["a", "lead", "plain", "trail"]
        assert fa.keys("predictions/a/b/a") == fb.keys("predictions/a/b/a") and fa.keys("predictions/a/b/a")[0].startswith("b-")
    if os.path.exists(H5DIFF):
        r = subprocess.run([H5DIFF, a, b], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


def test_a_rewritten_path_is_never_served_from_the_old_mapping(tmp_path, monkeypatch):
    """The reader caches (mapped scanner, libhdf5 handles) are keyed by path for the life of the process; a path that
    is rewritten -- truncated in place to a smaller file (touching the old mapping past the new end is a SIGBUS), or
    unlinked and recreated (the old inode would be served silently) -- must be noticed on the next call: every cache
    hit is checked against the path's device, inode, size and mtime."""
    rng = np.random.default_rng(5)
    p = str(tmp_path / "p_0.hdf")
    big = _batch(300, rng, contigs=("first",))
    small = _batch(5, rng, contigs=("second",))
    for mode in (None, "libhdf5"):
        if mode:
            monkeypatch.setenv("HELEN_IO_READER", mode)
        else:
            monkeypatch.delenv("HELEN_IO_READER", raising=False)
        _write(p, [big], monkeypatch, None)
        assert len(native_io.list_regions(p, "first")) == 24
        _write(p, [small], monkeypatch, None)                     # same inode, truncated: a smaller file
        assert native_io.list_regions(p, "first") in (None, [])
        assert len(native_io.list_regions(p, "second")) == 1
        os.unlink(p)                                              # new inode under the old name
        _write(str(tmp_path / "tmp.hdf"), [big], monkeypatch, None)
        os.rename(str(tmp_path / "tmp.hdf"), p)
        assert len(native_io.list_regions(p, "first")) == 24
        assert native_io.list_regions(p, "second") in (None, [])
        # rewritten by ANOTHER writer of the bytes (not this library): plain file copy over the path
        _write(str(tmp_path / "tmp.hdf"), [small], monkeypatch, None)
        with open(str(tmp_path / "tmp.hdf"), "rb") as src, open(p, "wb") as dst:
            dst.write(src.read())
        assert len(native_io.list_regions(p, "second")) == 1 and native_io.list_regions(p, "first") in (None, [])
        os.unlink(p)
    native_io.close_readers()
