"""Host-side rows of the path on CPU: HDF5 access, the image reader, the
prediction writer (8a9), checkpoint I/O (8a8), sharding (8a10) and the CLI surface."""
import os

import numpy as np
import pytest

from helen_amd import file_manager, hdf5, native_io
from helen_amd.data_store import DataStore
from helen_amd.sequence_dataset import SequenceDataset
from helen_amd.synthetic import write_image_dir, write_image_file
from helen_amd.weights import make_images, make_weights

pytestmark = pytest.mark.skipif(not hdf5.available(), reason="libhdf5 not loadable")


def test_hdf5_roundtrip(tmp_path):
    p = str(tmp_path / "t.h5")
    with hdf5.File(p, "w") as f:
        f.write("a/b/c/u8", np.arange(12, dtype=np.uint8).reshape(3, 4))
        f.write("a/b/i64", np.array([-5, 7], np.int64))
        f.write("a/scalar", 42)
        f.write("a/name", "chr20")
        f.write("z", np.zeros((0, 3), np.uint32))
        with pytest.raises(hdf5.Hdf5Error):
            f.write("a/scalar", 1)          # already exists
    with hdf5.File(p) as f:
        assert f.keys("/") == ["a", "z"] and f.keys("a") == ["b", "name", "scalar"]
        assert "a/b/c/u8" in f and "a/b/nope" not in f and "q/r" not in f
        u8 = f.read("a/b/c/u8")
        assert u8.dtype == np.uint8 and u8.tolist() == np.arange(12).reshape(3, 4).tolist()
        assert f.read("a/b/i64").tolist() == [-5, 7]
        assert f.read("a/b/i64", np.uint8).dtype == np.uint8
        s = f.read("a/scalar")
        assert s.shape == () and s.dtype == np.int64 and int(s) == 42
        assert f.read("a/name").reshape(-1)[0] == "chr20"
        assert f.read("z").shape == (0, 3)
        with pytest.raises(hdf5.Hdf5Error):
            f.read("missing")


def test_dataset_padding_and_order(tmp_path):
    img = make_images(5, seed=3)
    lengths = np.array([1000, 613, 1000, 1, 1000])
    p = str(tmp_path / "one.h5")
    write_image_file(p, img, lengths=lengths)
    ds = SequenceDataset(None, file_list=[p])
    assert len(ds) == 5
    # HDF5 key order is name order, as h5py's .keys()
    assert [n for _, n in ds.all_images] == sorted(n for _, n in ds.all_images)
    by_start = {}
    for i in range(5):
        contig, cs, ce, chunk, image, position, path = ds[i]
        assert contig == "chr20_synth" and ce == cs + 1000 and chunk == 0 and path == p
        assert image.shape == (1000, 90) and image.dtype == np.uint8
        assert position.shape == (1000, 3) and position.dtype == np.int64
        by_start[cs] = (image, position)
    for k, L in enumerate(lengths):
        image, position = by_start[800 * k]
        assert np.array_equal(image[:L], img[k, :L])
        assert not image[L:].any()                                  # zero rows
        assert (position[L:] == -1).all()                           # [-1,-1,-1] rows
        assert np.array_equal(position[:L, 0], 800 * k + np.arange(L))


def test_batches_sequential_short_last_and_workers(tmp_path):
    files = write_image_dir(str(tmp_path / "imgs"), 37, n_files=3, seed=5, short_every=5)
    ds = SequenceDataset(str(tmp_path / "imgs"))
    assert len(ds) == 37 and ds.num_batches(8) == 5
    a = list(ds.iter_batches(8))
    assert [b.images.shape[0] for b in a] == [8, 8, 8, 8, 5]       # drop_last=False
    assert a[0].contig_start.dtype == np.int64 and isinstance(a[0].contig[0], str)
    order = [(f, int(s)) for b in a for f, s in zip(b.filenames, b.contig_start)]
    want = [(p, int(n.split("-")[1])) for p, n in ds.all_images]
    assert order == want                                            # index order, no shuffle
    b = list(ds.iter_batches(8, num_workers=2))
    assert len(b) == len(a)
    for x, y in zip(a, b):
        assert np.array_equal(x.images, y.images) and np.array_equal(x.positions, y.positions)
        assert x.contig == y.contig and x.filenames == y.filenames
    assert files == file_manager.get_file_paths_from_directory(str(tmp_path / "imgs"))


def test_datastore_layout_dtypes_wrap_and_duplicates(tmp_path):
    out = str(tmp_path / "pred_0.hdf")
    pos = np.stack([np.arange(1000), np.zeros(1000), np.zeros(1000)], 1).astype(np.int64)
    pos[990:] = -1
    bases = (np.arange(1000) % 5).astype(np.int64)       # labels arrive as int64 from argmax
    rles = (np.arange(1000) % 11).astype(np.int64)
    with DataStore(out, "w") as s:
        s.write_prediction("ctg", np.int64(100), np.int64(1100), np.int64(0), pos, bases, rles, "f")
        s.write_prediction("ctg", np.int64(100), np.int64(1100), np.int64(1), pos, rles % 5, bases, "f")
        s.write_prediction("ctg", np.int64(100), np.int64(1100), np.int64(0), pos, bases * 0, rles * 0, "f")
        s.write_prediction("other", 5, 1005, 0, pos, bases, rles, "f")
    with hdf5.File(out) as f:
        assert f.keys("predictions") == ["ctg", "other"]
        assert f.keys("predictions/ctg") == ["ctg-100-1100"]
        root = "predictions/ctg/ctg-100-1100"
        assert f.keys(root) == ["0", "1", "contig_end", "contig_start"]
        assert int(f.read(root + "/contig_start")) == 100 and int(f.read(root + "/contig_end")) == 1100
        assert f.read(root + "/contig_start").shape == ()
        p = f.read(root + "/0/position")
        assert p.dtype == np.uint32 and p.shape == (1000, 3)
        assert (p[990:] == 4294967295).all() and p[5, 0] == 5       # -1 wraps (DataStore.py:126)
        b = f.read(root + "/0/bases")
        assert b.dtype == np.uint8 and np.array_equal(b, bases)     # duplicate did not overwrite
        assert f.read(root + "/0/rles").dtype == np.uint8
        assert np.array_equal(f.read(root + "/1/rles"), bases)


def test_round_robin_sharding():
    files = ["f%d" % i for i in range(7)]
    assert file_manager.shard_round_robin(files, 3) == [["f0", "f3", "f6"], ["f1", "f4"], ["f2", "f5"]]
    assert file_manager.shard_round_robin(files[:2], 8) == [["f0"], ["f1"]]   # empties dropped
    assert file_manager.shard_round_robin([], 4) == []
    chunks = file_manager.chunk_it(list(range(10)), 3)
    assert sum(chunks, []) == list(range(10))


def test_file_listing_filters_h5(tmp_path):
    for n in ("b.h5", "a.hdf5", "c.hdf", "d.txt", "e_h5"):
        (tmp_path / n).write_bytes(b"")
    got = [os.path.basename(p) for p in file_manager.get_file_paths_from_directory(str(tmp_path))]
    assert got == ["b.h5", "e_h5"]     # last two characters == 'h5' (CallConsensusInterface.py:43)


def test_checkpoint_roundtrip_and_module_prefix(tmp_path):
    import torch

    from helen_amd.model_handler import ModelHandler
    from helen_amd.transducer import TransducerGRU
    w = make_weights(seed=9)
    path = str(tmp_path / "m.pkl")
    ModelHandler.save_model(w, None, 128, 1, 3, path)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"model_state_dict", "model_optimizer", "hidden_size", "gru_layers", "epochs"}
    model, hidden, layers, epochs = ModelHandler.load_simple_model(path, 1, 90, 1000, 5, 11)
    assert (hidden, layers, epochs) == (128, 1, 3)
    sd = model.state_dict()
    assert all(np.array_equal(sd[k].numpy(), w[k]) for k in w)
    # a checkpoint saved from a DataParallel/DDP-wrapped model (ModelHander.py:70-75)
    torch.save({"model_state_dict": {"module." + k: torch.from_numpy(v) for k, v in w.items()},
                "model_optimizer": {}, "hidden_size": 128, "gru_layers": 1, "epochs": 0}, path)
    model2, _, _, _ = ModelHandler.load_simple_model(path, 1, 90, 1000, 5, 11)
    assert np.array_equal(model2.state_dict()["dense2_rle.bias"].numpy(), w["dense2_rle.bias"])
    m = TransducerGRU(1, 90, 1, 128, 5, 11)
    assert tuple(m.init_hidden(7, 1).shape) == (7, 2, 128)
    bad = dict(w)
    bad.pop("dense1_base.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    with pytest.raises(ValueError):
        TransducerGRU(1, 90, 2, 128, 5, 11)


def test_cli_flags_and_validation(tmp_path, capsys):
    from helen_amd import cli
    from helen_amd.call_consensus import call_consensus, plan_devices
    p = cli.build_parser()
    a = p.parse_args(["polish", "-i", "x", "-m", "y"])
    assert (a.batch_size, a.num_workers, a.threads, a.output_dir, a.output_prefix, a.gpu_mode,
            a.device_ids, a.callers) == (512, 8, 1, "./output/", "HELEN_prediction", False, None, 8)
    b = p.parse_args(["call_consensus", "-i", "x", "-m", "y", "-b", "256", "-g", "-d_ids", "0,2"])
    assert b.threads == 16 and b.gpu_mode and b.device_ids == "0,2" and b.batch_size == 256
    assert plan_devices(None, 4) == ([0, 1, 2, 3], 4)
    assert plan_devices("1,3", 4) == ([1, 3], 2)
    with pytest.raises(ValueError):
        plan_devices("5", 4)
    model = tmp_path / "m.pkl"
    model.write_bytes(b"x")
    for kwargs in (dict(model_path=str(tmp_path / "nope.pkl")), dict(image_dir=str(tmp_path / "nodir")),
                   dict(batch_size=0), dict(num_workers=-1), dict(threads=0), dict(gpu_mode=False)):
        args = dict(image_dir=str(tmp_path), model_path=str(model), batch_size=4, num_workers=0,
                    threads=1, output_dir=str(tmp_path / "out"), output_prefix="p", gpu_mode=True,
                    device_ids=None, callers=1)
        args.update(kwargs)
        with pytest.raises(SystemExit) as e:
            call_consensus(**args)
        assert e.value.code == 1
    assert cli.main(["version"]) == 0


def test_labeled_dataset_and_batch_losses(tmp_path):
    """The evaluation loader (models/dataloader.py:44-62) returns (image, label_base, label_run_length)
    as stored, refuses short images and out-of-range labels; batch_losses finishes the kernel's partial
    sums into nn.CrossEntropyLoss means per loader batch."""
    from helen_amd.evaluate import SequenceDataset, batch_losses
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images
    img = make_images(5, seed=3)
    rng = np.random.default_rng(1)
    lb = rng.integers(0, 5, (5, 1000), dtype=np.uint8)
    lr = rng.integers(0, 11, (5, 1000), dtype=np.uint8)
    d = tmp_path / "lab"
    d.mkdir()
    # first_window=100: image names then sort (HDF5 key order = string order) in window order
    write_image_file(str(d / "x.h5"), img, first_window=100, labels=(lb, lr))
    ds = SequenceDataset(str(d))
    assert len(ds) == 5
    image, b, r = ds[3]
    assert np.array_equal(image, img[3]) and np.array_equal(b, lb[3]) and np.array_equal(r, lr[3])
    images, bb, rr = ds.read_range(1, 4)
    assert images.shape == (3, 1000, 90) and np.array_equal(bb, lb[1:4]) and np.array_equal(rr, lr[1:4])
    # short image: this loader does not pad (torch's collate would fail on the ragged batch)
    s = tmp_path / "short"
    s.mkdir()
    write_image_file(str(s / "y.h5"), img[:1], lengths=np.array([613]), labels=(lb[:1], lr[:1]))
    with pytest.raises(ValueError, match="IMAGE SIZE ERROR"):
        SequenceDataset(str(s)).read_range(0, 1)
    bad = tmp_path / "bad"
    bad.mkdir()
    lb2 = lb.copy()
    lb2[0, 5] = 9
    write_image_file(str(bad / "z.h5"), img[:1], labels=(lb2[:1], lr[:1]))
    with pytest.raises(ValueError, match="LABEL OUT OF RANGE"):
        SequenceDataset(str(bad)).read_range(0, 1)
    # batch_losses: two loader batches (3 + 2 windows) from per-window partial sums
    stats = rng.random((5, 19, 10, 3)).astype(np.float32)
    loss_b, loss_r = batch_losses(stats, [3, 2])
    s64 = stats.astype(np.float64).sum(axis=2)
    np.testing.assert_allclose(loss_b[0], s64[:3, :, 0].sum(0) / 300)
    np.testing.assert_allclose(loss_r[1], s64[3:, :, 1].sum(0) / s64[3:, :, 2].sum(0))


def test_compressed_images_and_schema_check(tmp_path):
    """Chunked + deflated image / position datasets (what an h5py writer with compression='gzip' makes)
    read the same through both readers; check_images reports the storage facts and flags files the path
    cannot take."""
    import io

    from helen_amd.check_images import check_image_directory
    img = make_images(6, seed=9)
    good = tmp_path / "good"
    good.mkdir()
    write_image_file(str(good / "z.h5"), img, first_window=100, lengths=np.array([1000, 1000, 613, 1000, 1000, 1000]),
                     gzip=4)
    with hdf5.File(str(good / "z.h5")) as f:
        name = f.keys("images")[0]
        i = f.info("images/" + name + "/image")
        assert i["layout"] == "chunked" and i["filters"] == [1] and i["chunk"] == (256, 90) and i["class"] == "int"
    ds = SequenceDataset(str(good))
    assert len(ds) == 6
    from helen_amd.sequence_dataset import _load_batch
    batch = _load_batch(ds.all_images)                    # native reader when libhelen_io.so is built
    assert np.array_equal(batch.images[0], img[0]) and np.array_equal(batch.images[2][:613], img[2][:613])
    assert not batch.images[2][613:].any() and (batch.positions[2][613:] == -1).all()
    item = ds[3]                                          # ctypes reader
    assert np.array_equal(item[4], img[3])
    rep = io.StringIO()
    assert check_image_directory(str(good), out=rep) == 0
    assert "deflate" in rep.getvalue() and "6 images" in rep.getvalue()
    bad = tmp_path / "bad"
    bad.mkdir()
    with hdf5.File(str(bad / "b.h5"), "w") as f:
        base = "images/x-0-1000-0/"
        f.write(base + "contig", "x")
        for k in ("contig_start", "contig_end"):
            f.write(base + k, np.array([0], np.int64))
        f.write(base + "image", np.zeros((1000, 10), np.uint8))       # wrong width, no feature_chunk_idx
        f.write(base + "position", np.zeros((900, 3), np.int64))      # row count differs
    rep = io.StringIO()
    n = check_image_directory(str(bad), out=rep)
    text = rep.getvalue()
    assert n >= 2 and "missing dataset 'feature_chunk_idx'" in text and "expected integers [l <= 1000, 90]" in text


def _write_schema_variants(tmp_path):
    """One image file holding eight storage variants of the reader's six datasets.  -> (path, variants, images)"""
    img = make_images(8, seed=33)
    rng = np.random.default_rng(3)
    variants = [
        dict(contig="fixed", ints=np.int64, scalar=False, image=np.uint8, position=np.int64, store={}),
        dict(contig="vlen", ints=np.int32, scalar=False, image=np.uint8, position=np.int32, store={}),
        dict(contig="scalar", ints=np.uint32, scalar=True, image=np.uint16, position=np.uint32, store={}),
        dict(contig="vlen_scalar", ints=np.int16, scalar=True, image=np.int32, position=np.uint64,
             store=dict(gzip=6, shuffle=True)),
        dict(contig="fixed", ints=np.uint64, scalar=False, image=np.int64, position=np.int64,
             store=dict(gzip=1)),
        dict(contig="vlen", ints=np.uint8, scalar=True, image=np.float32, position=np.int64,
             store=dict(shuffle=True)),
        dict(contig="fixed", ints=np.int8, scalar=False, image=np.uint8, position=np.int16, store={}, labels=True),
        dict(contig="scalar", ints=np.uint16, scalar=False, image=np.float64, position=np.int32,
             store=dict(gzip=9, shuffle=True), labels=True),
    ]
    path = str(tmp_path / "variants.h5")
    lengths = [1000, 1000, 613, 1000, 1, 999, 1000, 500]
    with hdf5.File(path, "w") as f:
        for i, v in enumerate(variants):
            L = lengths[i]
            small = np.dtype(v["ints"]).itemsize == 1            # one-byte integers cannot hold 800
            start, chunk = (100 if small else 800 * i), (i % 3)
            base = "images/w%02d/" % i
            f.write(base + "contig", "chr%d_variant" % i, string=v["contig"])
            for name, val in (("contig_start", start), ("contig_end", start + (27 if small else 1000)),
                              ("feature_chunk_idx", chunk)):
                f.write(base + name, np.array(val if v["scalar"] else [val], v["ints"]))
            f.write(base + "image", img[i, :L].astype(v["image"]), v["image"],
                    chunks=(min(L, 256), 90) if v["store"] else None, **v["store"])
            pos = np.stack([5000 + np.arange(L), rng.integers(0, 3, L), rng.integers(0, 2, L)], 1)
            f.write(base + "position", pos.astype(v["position"]), v["position"],
                    chunks=(L, 3) if v["store"] else None, **v["store"])
            if v.get("labels"):
                f.write(base + "label_base", np.zeros(L, np.uint8))
                f.write(base + "label_run_length", np.zeros(L, np.uint8))
            v["want"] = ("chr%d_variant" % i, start, start + (27 if small else 1000), chunk, L, pos)
    return path, variants, img


def test_reader_takes_every_plausible_schema_variant(tmp_path):
    """SURVEY.md 8f-2 without real data: whatever h5py or MarginPolish could plausibly have written for the six
    datasets the reference reader touches (dataloader_predict.py:64-70) must read the same through the native
    batch reader and the per-item reader: `contig` as a fixed-length, variable-length or scalar string; the three
    integers in any width, signed or not, as [1] arrays or scalars; `image` as uint8, a wider integer or a
    float; `position` as int32 / int64 / uint32 / uint64 (the reader casts it to int); chunked + shuffle +
    deflate storage; label datasets present in an inference directory."""
    from helen_amd.sequence_dataset import _load_batch
    path, variants, img = _write_schema_variants(tmp_path)
    ds = SequenceDataset(None, file_list=[path])
    assert len(ds) == 8
    batch = _load_batch(ds.all_images)                            # native reader when libhelen_io.so is built
    for i, v in enumerate(variants):
        contig, start, end, chunk, L, pos = v["want"]
        item = ds[i]                                              # per-item (ctypes) reader
        for got in ((batch.contig[i], batch.contig_start[i], batch.contig_end[i], batch.chunk_id[i],
                     batch.images[i], batch.positions[i]), item[:6]):
            assert (got[0], int(got[1]), int(got[2]), int(got[3])) == (contig, start, end, chunk), (i, got[:4])
            assert got[4].dtype == np.uint8 and np.array_equal(got[4][:L], img[i, :L]) and not got[4][L:].any()
            assert np.array_equal(got[5][:L], pos) and (got[5][L:] == -1).all()


def test_call_consensus_vets_the_image_directory_first(tmp_path, capsys):
    """A directory with a malformed image is refused before any process is started, with the reference reader's
    IMAGE SIZE ERROR text (dataloader_predict.py:85-86)."""
    from helen_amd.call_consensus import vet_image_directory
    from helen_amd.synthetic import write_image_dir
    good = str(tmp_path / "good")
    write_image_dir(good, 12, n_files=2, short_every=5)
    assert vet_image_directory(good) == 0
    bad = tmp_path / "bad"
    bad.mkdir()
    with hdf5.File(str(bad / "b.h5"), "w") as f:
        base = "images/x-0-1000-0/"
        f.write(base + "contig", "x")
        for k in ("contig_start", "contig_end", "feature_chunk_idx"):
            f.write(base + k, np.array([0], np.int64))
        f.write(base + "image", np.zeros((1000, 10), np.uint8))       # F = 10 is not this model's 90
        f.write(base + "position", np.zeros((1000, 3), np.int64))
    with pytest.raises(ValueError, match="IMAGE SIZE ERROR"):
        vet_image_directory(str(bad))
    assert "expected integers" in capsys.readouterr().err


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LABELED_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from helen_amd import native_io
from helen_amd.evaluate import SequenceDataset
ds = SequenceDataset(%(dir)r)
images, lb, lr = ds.read_range(0, len(ds))
np.savez(%(out)r, images=images, lb=lb, lr=lr, counts=np.array(native_io.reader_counts()))
'''


def test_labeled_images_native_reader_equals_the_per_item_reader(tmp_path):
    """The evaluation loader (models/dataloader.py:48-61) through helen_io_read_labeled -- scanner (a contiguous and a
    deflated file) and libhdf5 alone -- against the per-item ctypes reader; a short labeled image is the loader's
    IMAGE SIZE ERROR, not padded."""
    import subprocess
    import sys
    from helen_amd.evaluate import SequenceDataset
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images
    rng = np.random.default_rng(4)
    img = make_images(10, seed=8)
    lb = rng.integers(0, 5, (10, 1000), dtype=np.uint8)
    lr = rng.integers(0, 11, (10, 1000), dtype=np.uint8)
    d = tmp_path / "labeled"
    d.mkdir()
    write_image_file(str(d / "a.h5"), img[:6], first_window=0, labels=(lb[:6], lr[:6]))
    write_image_file(str(d / "b.h5"), img[6:], first_window=6, labels=(lb[6:], lr[6:]), gzip=4)   # inflated by the scanner
    ds = SequenceDataset(str(d))
    order = [int(name.split("-")[1]) // 800 for _, name in ds.all_images]
    want = [np.stack([ds[i][k] for i in range(len(ds))]) for k in range(3)]          # per-item reader
    assert np.array_equal(want[0], img[order]) and np.array_equal(want[1], lb[order]) and np.array_equal(want[2], lr[order])
    got = {}
    for reader in ("", "libhdf5"):
        env = dict(os.environ)
        env.pop("HELEN_IO_READER", None)
        if reader:
            env["HELEN_IO_READER"] = reader
        out = str(tmp_path / ("l_%s.npz" % (reader or "default")))
        r = subprocess.run([sys.executable, "-c", LABELED_CHILD % {"root": ROOT, "dir": str(d), "out": out}], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[reader] = dict(np.load(out))
        for k, w in zip(("images", "lb", "lr"), want):
            assert np.array_equal(got[reader][k], w), (reader, k)
    assert tuple(got[""]["counts"]) == (10, 0) and tuple(got["libhdf5"]["counts"]) == (0, 10)
    short = tmp_path / "short"
    short.mkdir()
    write_image_file(str(short / "s.h5"), img[:2], lengths=[1000, 999], labels=(lb[:2], lr[:2]))
    ds = SequenceDataset(str(short))
    with pytest.raises(ValueError, match="IMAGE SIZE ERROR"):
        ds.read_range(0, 2)


def _load_generator(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "golden", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # only its input builders are used here: nothing of the reference
    return mod


def _io_fixture():
    import gzip
    import json
    with gzip.open(os.path.join(ROOT, "tests", "golden", "io_ref.json.gz"), "rt") as f:
        return json.load(f)


READER_CHILD = r'''
import json, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests/golden")
from make_golden_io import digest
from helen_amd.sequence_dataset import SequenceDataset, _load_batch
ds = SequenceDataset(None, file_list=%(paths)r)
b = _load_batch(ds.all_images)
print(json.dumps([{"name": n, "contig": b.contig[k], "contig_start": int(b.contig_start[k]), "contig_end": int(b.contig_end[k]),
                   "chunk_id": int(b.chunk_id[k]), "image": digest(b.images[k]), "position": digest(b.positions[k])}
                  for k, (_, n) in enumerate(ds.all_images)]))
'''


def test_reader_equals_the_reference_reader_itself(tmp_path):
    """tests/golden/io_ref.json.gz holds what the REFERENCE's own SequenceDataset (dataloader_predict.py:18-95), imported
    from the reference tree and run on the image files of make_golden_io.io_case_files, returned item by item.  This
    package's per-item reader and its native batch reader (scanner, libhdf5, either alone) must return the same
    order, names, contig strings (quotes stripped), integers, padded images and positions."""
    import json
    import subprocess
    import sys
    from helen_amd.sequence_dataset import SequenceDataset
    gen = _load_generator("make_golden_io")
    want = _io_fixture()["reader"]
    paths = gen.io_case_files(str(tmp_path))
    ds = SequenceDataset(None, file_list=paths)
    assert [(os.path.basename(p), n) for p, n in ds.all_images] == [(w["file"], w["name"]) for w in want]
    for k, w in enumerate(want):
        contig, start, end, chunk, image, position, path = ds[k]
        assert (str(contig), int(start), int(end), int(chunk), os.path.basename(path)) == \
            (w["contig"], w["contig_start"], w["contig_end"], w["chunk_id"], w["returned_file"]), k
        assert gen.digest(image) == w["image"] and gen.digest(np.asarray(position, np.int64)) == w["position"], k
    for reader in ("", "libhdf5", "direct"):
        env = dict(os.environ)
        env.pop("HELEN_IO_READER", None)
        if reader:
            env["HELEN_IO_READER"] = reader
        r = subprocess.run([sys.executable, "-c", READER_CHILD % {"root": ROOT, "paths": paths}], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got = json.loads(r.stdout.strip().splitlines()[-1])
        for g, w in zip(got, want):
            for key in ("name", "contig", "contig_start", "contig_end", "chunk_id", "image", "position"):
                assert g[key] == w[key], (reader, key, g["name"])


@pytest.mark.parametrize("writer", [None, "libhdf5"])
def test_writer_equals_the_reference_writer_itself(tmp_path, monkeypatch, writer):
    """... and what the REFERENCE's own DataStore.write_prediction (DataStore.py:83-133) stored for the windows of
    make_golden_io.writer_case: the same tree of dataset names, dtypes (position uint32 with wrapped -1 padding,
    labels uint8, scalar int64 bounds), shapes and bytes, the repeated window skipped -- from either writer here."""
    from helen_amd.data_store import DataStore
    if writer:
        monkeypatch.setenv("HELEN_IO_WRITER", writer)
    else:
        monkeypatch.delenv("HELEN_IO_WRITER", raising=False)
    gen = _load_generator("make_golden_io")
    want = _io_fixture()["writer"]
    path = str(tmp_path / "pred.hdf")
    store = DataStore(path, mode="w")
    for contig, start, end, chunk, pos, bases, rles in gen.writer_case():
        store.write_prediction(contig, np.int64(start), np.int64(end), np.int64(chunk), pos, bases, rles, "x.h5")
    store.close()
    with hdf5.File(path, "r") as f:
        got = gen.walk(f)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k


def _strict_check_cases(tmp_path):
    """Directories for check_images --strict: plain files (direct emitter and libhdf5 writer), deflated storage,
    every schema variant, and one the reader must refuse."""
    import io
    import json

    from helen_amd.check_images import main as check_main
    from helen_amd.synthetic import write_image_dir
    plain = str(tmp_path / "plain")
    write_image_dir(plain, 40, n_files=2, seed=5, short_every=7, direct=True)
    write_image_file(os.path.join(plain, "lib.h5"), make_images(5, seed=6), first_window=500)
    rep = str(tmp_path / "plain.json")
    out = io.StringIO()
    assert check_main(plain, strict=True, json_path=rep, out=out) == 0, out.getvalue()
    r = json.load(open(rep))
    assert r["verdict"] == "ready" and r["images"] == r["images_inspected"] == r["images_read"] == 45
    assert [f["reader_path"] for f in r["files"]] == ["direct scanner"] * 3
    assert r["files"][0]["datasets"]["image"]["types"] == ["uint8"] and r["files"][1]["datasets"]["image"]["rows_min"] == 613
    # deflated storage: the scanner inflates it itself (round 4), the report names the storage class and its price
    z = tmp_path / "deflated"
    z.mkdir()
    write_image_file(str(z / "z.h5"), make_images(6, seed=9), first_window=100, gzip=4)
    out = io.StringIO()
    assert check_main(str(z), strict=True, json_path=rep, out=out) == 0, out.getvalue()
    r = json.load(open(rep))
    assert r["files"][0]["reader_path"] == "direct scanner" and r["files"][0]["datasets"]["image"]["filters"] == [["deflate"]]
    assert r["files"][0]["storage_class"] == "deflate" and "stored deflate" in out.getvalue()
    # a chunk index the scanner does not walk (paged fixed array): read through libhdf5, and the report says so
    pg = tmp_path / "paged"
    pg.mkdir()
    write_image_file(str(pg / "p.h5"), make_images(2, seed=9), first_window=100, libver="latest", chunks=(1, 45))
    out = io.StringIO()
    assert check_main(str(pg), strict=True, json_path=rep, out=out) == 0, out.getvalue()
    r = json.load(open(rep))
    assert r["files"][0]["reader_path"] == "libhdf5" and r["files"][0]["storage_class"] == "libhdf5"
    assert "read through the libhdf5" in out.getvalue()
    # every schema variant in one file: all eight images are taken
    v = tmp_path / "variants"
    v.mkdir()
    _write_schema_variants(v)
    out = io.StringIO()
    assert check_main(str(v), strict=True, json_path=rep, out=out) == 0, out.getvalue()
    r = json.load(open(rep))
    assert r["images_read"] == 8 and sorted(r["files"][0]["datasets"]["position"]["types"]) == [
        "int16", "int32", "int64", "uint32", "uint64"]
    # an image the reader refuses: exit code 2, the refusal carries the reader's own text
    bad = tmp_path / "bad"
    bad.mkdir()
    write_image_file(str(bad / "ok.h5"), make_images(3, seed=1))
    with hdf5.File(str(bad / "wide.h5"), "w") as f:
        base = "images/x-0-1000-0/"
        f.write(base + "contig", "x")
        for k in ("contig_start", "contig_end", "feature_chunk_idx"):
            f.write(base + k, np.array([0], np.int64))
        f.write(base + "image", np.zeros((1000, 10), np.uint8))
        f.write(base + "position", np.zeros((1000, 3), np.int64))
    out = io.StringIO()
    assert check_main(str(bad), strict=True, json_path="-", out=out) == 2
    assert "REFUSED x-0-1000-0" in out.getvalue() and "IMAGE SIZE ERROR" in out.getvalue()
    # without --strict / --json the command is the sampled report it was
    out = io.StringIO()
    assert check_main(plain, out=out) == 0 and "0 problem(s)" in out.getvalue()


def test_check_images_strict_and_json(tmp_path):
    """`python -m helen_amd check_images -i <dir> --strict --json report.json`: the one command a user with real
    MarginPolish output runs before the first polish."""
    _strict_check_cases(tmp_path)


@pytest.mark.gpu
def test_check_images_strict_on_the_gpu_box(tmp_path):
    """The same on the GPU box's host (its libhdf5, its file system), through the command line."""
    import subprocess
    import sys
    _strict_check_cases(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "helen_amd", "check_images", "-i", str(tmp_path / "plain"), "--strict",
                        "--json", str(tmp_path / "cli.json")], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "ready" in r.stdout, r.stdout + r.stderr


def test_reader_threads_equal_the_per_name_reader(tmp_path):
    """helen_io_read_image_runs -- images addressed by (file, position in name order), read by several native threads --
    against the per-name batch reader (which the reference's own reader pins, test_*_equals_the_reference_*): runs that
    start and end inside files, a file with short images, 1 / 3 / 8 threads, a piece boundary (32 images) inside a run."""
    from helen_amd import native_io
    from helen_amd.sequence_dataset import SequenceDataset, _load_batch
    from helen_amd.synthetic import write_image_dir
    if not native_io.available():
        pytest.skip("libhelen_io.so not built")
    files = write_image_dir(str(tmp_path / "img"), 150, n_files=3, short_every=7)
    ds = SequenceDataset(None, file_list=files)
    assert [(p, n) for p, n, _ in ds.runs] == [(f, 50) for f in files] and not any(lib for _, _, lib in ds.runs)
    assert len(ds) == 150
    pairs = ds.all_images
    assert [n for _, n in pairs[:50]] == native_io.image_names(files[0], 0, 50) == native_io.list_images(files[0])
    runs = [(files[0], 13, 37), (files[1], 0, 50), (files[2], 0, 41)]
    want = _load_batch(pairs[13:141])
    for threads in (1, 3, 8):
        n = 128
        images = np.full((n, 1000, 90), 7, np.uint8)
        positions = np.full((n, 1000, 3), 7, np.int64)
        meta = np.full((n, 3), 7, np.int64)
        contigs = np.full((n, native_io.NAME_BYTES), 7, np.uint8)
        assert native_io.read_image_runs(runs, threads, images, positions, meta, contigs) == 0
        assert np.array_equal(images, want.images) and np.array_equal(positions, want.positions)
        assert np.array_equal(meta[:, 0], want.contig_start) and np.array_equal(meta[:, 2], want.chunk_id)
        assert native_io.contig_names(contigs) == want.contig
    assert ds.call_runs(64) == [[(files[0], 0, 50), (files[1], 0, 14)], [(files[1], 14, 36), (files[2], 0, 28)],
                                [(files[2], 28, 22)]]
    with pytest.raises(IOError, match="out of bounds"):
        native_io.read_image_runs([(files[0], 40, 11)], 2, images, positions, meta, contigs)
    # a rewritten file is noticed (identity check), a forgotten one is indexed again
    native_io.forget_images(files[0])
    assert native_io.index_images(files[0]) == (50, False)
    write_image_dir(str(tmp_path / "img"), 30, n_files=3)
    assert native_io.index_images(files[0]) == (10, False)


def test_reader_threads_take_library_files_too(tmp_path, monkeypatch):
    """A file the direct scanner does not take is read through libhdf5 by the same entry point (serialised), the count of
    such images is returned, and the values are the per-name reader's."""
    from helen_amd import native_io
    from helen_amd.sequence_dataset import SequenceDataset, _load_batch
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images
    if not native_io.available():
        pytest.skip("libhelen_io.so not built")
    monkeypatch.setenv("HELEN_IO_READER", "libhdf5")         # (read once per process: this test runs in a child)
    import subprocess
    import sys
    img = make_images(40, seed=11)
    a, b = str(tmp_path / "a.h5"), str(tmp_path / "b.h5")
    write_image_file(a, img[:24], first_window=0)
    write_image_file(b, img[24:], first_window=24)
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from helen_amd import native_io
from helen_amd.sequence_dataset import SequenceDataset, _load_batch
ds = SequenceDataset(None, file_list=[%r, %r])
assert [(n, lib) for _, n, lib in ds.runs] == [(24, True), (16, True)], ds.runs
want = _load_batch(ds.all_images)
n = 40
images = np.zeros((n, 1000, 90), np.uint8); positions = np.zeros((n, 1000, 3), np.int64)
meta = np.zeros((n, 3), np.int64); contigs = np.zeros((n, native_io.NAME_BYTES), np.uint8)
assert native_io.read_image_runs([(%r, 0, 24), (%r, 0, 16)], 4, images, positions, meta, contigs) == 40
assert np.array_equal(images, want.images) and np.array_equal(positions, want.positions)
assert native_io.reader_counts()[1] >= 40
print("ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), a, b, a, b)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


H5DUMP = "/opt/conda/bin/h5dump"


def _h5dump_dataset(path, name):
    """(type text, shape or None for a scalar, values) of one dataset as libhdf5's own `h5dump` prints it: a reader that
    shares no code with helen_amd/hdf5.py or the scanner."""
    import re
    import subprocess
    r = subprocess.run([H5DUMP, "-w", "0", "-d", name, path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    text = r.stdout
    m = re.search(r"DATATYPE\s+(H5T_STRING\s*\{[^}]*\}|\S+)", text)
    dtype = re.sub(r"\s+", " ", m.group(1))
    m = re.search(r"DATASPACE\s+SIMPLE\s*\{\s*\(([^)]*)\)", text)
    shape = tuple(int(v) for v in m.group(1).split(",")) if m else None
    body = text[text.index("DATA {") + 6:text.rindex("}")]
    body = body[:body.rindex("}")] if body.strip().endswith("}") else body
    values = []
    for line in body.splitlines():
        line = line.strip()
        if not line or line == "}":
            continue
        line = re.sub(r"^\([\d,]*\):\s*", "", line).rstrip(",")
        if line.startswith('"'):
            values.extend(re.findall(r'"((?:[^"\\]|\\.)*)"', line))
        else:
            values.extend(float(v) if ("." in v or "e" in v.lower() or "inf" in v or "nan" in v) else int(v)
                          for v in (t.strip() for t in line.split(",")) if v)
    return dtype, shape, values


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="h5dump not installed")
def test_hdf5_layer_against_libhdf5s_own_tool(tmp_path):
    """The golden generators run the reference's reader / writer / stitch over an h5py stand-in built on helen_amd/hdf5.py
    (tests/golden/reference_env.py), so that binding is on both sides of those comparisons.  Here it -- and the direct
    scanner and emitter -- are put against `h5dump`, libhdf5's own command-line reader, which shares no code with any of
    them: (a) an image file with every schema variant the reference's reader takes, written through hdf5.py: types, shapes
    and values as h5dump prints them = what hdf5.py reads back = what the scanner reads; (b) a prediction file from the
    direct emitter and one from the libhdf5 writer: names, types (uint32 positions with the -1 wrap, uint8 labels, int64
    bounds) and values as h5dump prints them."""
    from helen_amd.data_store import DataStore
    from helen_amd.sequence_dataset import SequenceDataset, _load_batch
    img = make_images(3, seed=13, mode="pileup")
    path = str(tmp_path / "variants.h5")
    rng = np.random.default_rng(3)
    want = {}
    with hdf5.File(path, "w") as f:
        for i in range(3):
            base = "images/w%d/" % i
            L = [1000, 37, 1][i]
            contig = ["chr1", "chr2'q", "c3"][i]
            f.write(base + "contig", contig, string=["fixed", "vlen", "scalar"][i])
            ints = [np.int64, np.int32, np.uint16][i]
            for name, val in (("contig_start", 800 * i), ("contig_end", 800 * i + 1000), ("feature_chunk_idx", i)):
                f.write(base + name, np.array(val if i == 1 else [val], ints))
            it = [np.uint8, np.uint16, np.int32][i]
            f.write(base + "image", img[i, :L].astype(it), it, chunks=(64, 90) if i == 1 else None, gzip=4 if i == 1 else None)
            pt = [np.int64, np.int32, np.uint32][i]
            pos = np.stack([100 + np.arange(L), rng.integers(0, 3, L), rng.integers(0, 2, L)], 1).astype(pt)
            f.write(base + "position", pos, pt)
            want[i] = (contig, ints, it, pt, L, pos)
    names = {np.int64: "H5T_STD_I64LE", np.int32: "H5T_STD_I32LE", np.uint16: "H5T_STD_U16LE", np.uint8: "H5T_STD_U8LE",
             np.uint32: "H5T_STD_U32LE"}
    with hdf5.File(path) as f:
        for i in range(3):
            contig, ints, it, pt, L, pos = want[i]
            base = "/images/w%d/" % i
            t, shape, vals = _h5dump_dataset(path, base + "contig")
            assert t.startswith("H5T_STRING") and vals == [contig]
            assert ("H5T_VARIABLE" in t) == (i == 1) and shape == (None if i == 2 else (1,))
            assert str(np.asarray(f.read(base + "contig")).reshape(-1)[0]) in (contig, "b'%s'" % contig) or \
                np.asarray(f.read(base + "contig")).reshape(-1)[0] in (contig, contig.encode())
            t, shape, vals = _h5dump_dataset(path, base + "contig_start")
            assert t == names[ints] and vals == [800 * i] and shape == (None if i == 1 else (1,))
            assert int(np.asarray(f.read(base + "contig_start")).reshape(-1)[0]) == 800 * i
            t, shape, vals = _h5dump_dataset(path, base + "image")
            assert t == names[it] and shape == (L, 90)
            assert np.array_equal(np.array(vals).reshape(L, 90), img[i, :L]) and np.array_equal(f.read(base + "image"), img[i, :L])
            t, shape, vals = _h5dump_dataset(path, base + "position")
            assert t == names[pt] and shape == (L, 3) and np.array_equal(np.array(vals).reshape(L, 3), pos)
            assert np.array_equal(f.read(base + "position"), pos)
    # the product's reader (direct scanner) on the same file returns what h5dump printed
    b = _load_batch(SequenceDataset(None, file_list=[path]).all_images)
    for i in range(3):
        contig, ints, it, pt, L, pos = want[i]
        assert np.array_equal(b.images[i, :L], img[i, :L]) and not b.images[i, L:].any()
        assert np.array_equal(b.positions[i, :L], pos.astype(np.int64)) and (b.positions[i, L:] == -1).all()
        assert (int(b.contig_start[i]), int(b.contig_end[i]), int(b.chunk_id[i])) == (800 * i, 800 * i + 1000, i)
    assert b.contig == ["chr1", '"chr2q"', "c3"]          # (the reference's array2string quirk: tests above)
    # prediction files of both writers
    P = np.full((1000, 3), -1, np.int64)
    P[:700, 0] = 5000 + np.arange(700)
    P[:700, 1:] = rng.integers(0, 3, (700, 2))
    B = rng.integers(0, 5, 1000).astype(np.uint8)
    R = rng.integers(0, 11, 1000).astype(np.uint8)
    for writer in ("", "libhdf5"):
        out = str(tmp_path / ("pred_%s.hdf" % (writer or "direct")))
        if writer:
            os.environ["HELEN_IO_WRITER"] = writer
        try:
            with DataStore(out, "w") as s:
                s.write_prediction("chrP", 5000, 6000, 2, P, B, R)
        finally:
            os.environ.pop("HELEN_IO_WRITER", None)
        root = "/predictions/chrP/chrP-5000-6000"
        assert _h5dump_dataset(out, root + "/contig_start") == ("H5T_STD_I64LE", None, [5000])
        assert _h5dump_dataset(out, root + "/contig_end") == ("H5T_STD_I64LE", None, [6000])
        t, shape, vals = _h5dump_dataset(out, root + "/2/position")
        assert t == "H5T_STD_U32LE" and shape == (1000, 3)
        got = np.array(vals, dtype=np.uint64).reshape(1000, 3)
        assert np.array_equal(got[:700], P[:700]) and (got[700:] == 4294967295).all()       # -1 wraps (DataStore.py:126)
        t, shape, vals = _h5dump_dataset(out, root + "/2/bases")
        assert t == "H5T_STD_U8LE" and shape == (1000,) and np.array_equal(np.array(vals), B)
        t, shape, vals = _h5dump_dataset(out, root + "/2/rles")
        assert t == "H5T_STD_U8LE" and np.array_equal(np.array(vals), R)


def test_writer_threads_give_the_same_file(tmp_path, monkeypatch):
    """The prediction writer reserves a window's block on its own thread and fills it -- the block's bytes, 3000 positions
    int64 -> uint32, two label rows -- on $HELEN_IO_WRITER_THREADS threads, a flush-full at a time (helen_amd/csrc/io.cpp:
    FillPool).  Any thread count gives the same FILE, byte for byte: many calls of 9 .. 700 windows (hundreds of rounds of
    the pool, buffers handed on in the middle of a call), duplicates and -1 padding rows included."""
    import hashlib
    rng = np.random.default_rng(7)
    n = 9000
    names = ["ctg%d" % (i // 2500) for i in range(n)]
    meta = np.zeros((n, 3), np.int64)
    region = np.arange(n) // 3
    meta[:, 0], meta[:, 1], meta[:, 2] = region * 2400, region * 2400 + 2400, np.arange(n) % 3
    meta[4000:4010] = meta[3990:4000]                       # repeats: skipped silently (DataStore.py:123)
    names[4000:4010] = names[3990:4000]
    positions = rng.integers(0, 1 << 40, (n, 1000, 3), dtype=np.int64)
    positions[::7, 900:] = -1                               # padding rows: 4294967295 in the file
    bases = rng.integers(0, 5, (n, 1000), dtype=np.uint8)
    rles = rng.integers(0, 11, (n, 1000), dtype=np.uint8)
    cuts = [0]
    while cuts[-1] < n:
        cuts.append(min(n, cuts[-1] + int(rng.choice([9, 33, 250, 700]))))
    digests = {}
    for threads in ("1", "2", "4", "7"):
        monkeypatch.setenv("HELEN_IO_WRITER_THREADS", threads)
        path = str(tmp_path / ("p_%s.hdf" % threads))
        w = native_io.Writer(path)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            w.write(native_io.pack_contigs(names[lo:hi]), meta[lo:hi], positions[lo:hi], bases[lo:hi], rles[lo:hi])
        w.close()
        digests[threads] = hashlib.sha1(open(path, "rb").read()).hexdigest()
    assert len(set(digests.values())) == 1, digests
    with hdf5.File(str(tmp_path / "p_4.hdf")) as f:
        i = 2501
        got = f.read("predictions/ctg1/ctg1-%d-%d/%d/position" % (meta[i, 0], meta[i, 1], meta[i, 2]))
        assert got.dtype == np.uint32 and np.array_equal(got, positions[i].astype(np.uint32))
        assert np.array_equal(f.read("predictions/ctg0/ctg0-0-2400/2/rles"), rles[2])
