"""End-to-end on the GPU box: synthetic MarginPolish image directory -> call_consensus / the CLI ->
prediction HDF5, checked against the CPU oracle window by window.  pytest -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helen_amd import hdf5
from helen_amd.weights import make_weights

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not hdf5.available(), reason="libhdf5 not loadable")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected(image_dir, weights):
    """Oracle labels for every image of the directory, keyed by (contig_start, chunk_id)."""
    import oracle
    from helen_amd.sequence_dataset import SequenceDataset
    ds = SequenceDataset(image_dir)
    items = [ds[i] for i in range(len(ds))]
    o = oracle.polish_batch(weights, np.stack([it[4] for it in items]))
    return {(it[1], it[3]): (o["bases"][i], o["rles"][i], it[5]) for i, it in enumerate(items)}


def _check_prediction_files(paths, expected):
    seen = 0
    for p in paths:
        with hdf5.File(p) as f:
            for contig in f.keys("predictions"):
                for prefix in f.keys("predictions/" + contig):
                    root = "predictions/%s/%s" % (contig, prefix)
                    cs = int(f.read(root + "/contig_start"))
                    assert prefix == "%s-%d-%d" % (contig, cs, int(f.read(root + "/contig_end")))
                    for chunk in f.keys(root):
                        if chunk in ("contig_start", "contig_end"):
                            continue
                        eb, er, pos = expected[(cs, int(chunk))]
                        b = f.read(root + "/" + chunk + "/bases")
                        r = f.read(root + "/" + chunk + "/rles")
                        q = f.read(root + "/" + chunk + "/position")
                        assert b.dtype == np.uint8 and r.dtype == np.uint8 and q.dtype == np.uint32
                        assert np.array_equal(b, eb) and np.array_equal(r, er)
                        assert np.array_equal(q, pos.astype(np.uint32))
                        seen += 1
    assert seen == len(expected)


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_dir
    d = tmp_path_factory.mktemp("pipe")
    w = make_weights(seed=11, input_scale=1.0 / 64.0)   # seed 11: labels of several classes (seed 31 calls every base "gap")
    model = str(d / "synthetic_model.pkl")
    ModelHandler.save_model(w, None, 128, 1, 0, model)
    img_dir = str(d / "images")
    write_image_dir(img_dir, 70, n_files=3, seed=77, short_every=9)
    return d, w, model, img_dir, _expected(img_dir, w)


def test_call_consensus_matches_oracle(workdir):
    from helen_amd.call_consensus import call_consensus
    d, w, model, img_dir, expected = workdir
    out = str(d / "out_api")
    call_consensus(img_dir, model, 16, 0, 1, out, "pred", True, "0", 1)
    files = [os.path.join(out, f) for f in sorted(os.listdir(out))]
    assert [os.path.basename(f) for f in files] == ["pred_0.hdf"]      # <prefix>_<rank>.hdf
    _check_prediction_files(files, expected)


def test_cli_polish_with_workers(workdir):
    d, w, model, img_dir, expected = workdir
    out = str(d / "out_cli")
    r = subprocess.run([sys.executable, "-m", "helen_amd", "polish", "-i", img_dir, "-m", model, "-b", "8",
                        "-w", "2", "-o", out, "-p", "asm", "-g", "-d_ids", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    pred_dirs = [x for x in os.listdir(out) if x.startswith("predictions_")]
    assert len(pred_dirs) == 1                                          # PolishInterface.py:65-69
    pdir = os.path.join(out, pred_dirs[0])
    files = [os.path.join(pdir, f) for f in sorted(os.listdir(pdir))]
    # -w 2: two reader workers; ONE prediction file per rank, as the reference writes it (predict_gpu.py:55)
    assert [os.path.basename(f) for f in files] == ["asm_0.hdf"]
    _check_prediction_files(files, expected)
    # ... and the stitched FASTA: one record, identical to stitching the same predictions again
    fasta = open(os.path.join(out, "asm.fa")).read().split("\n")
    assert fasta[0] == ">chr20_synth" and len(fasta[1]) > 1000 and set(fasta[1]) <= set("ACGTN"), fasta[0][:50]
    from helen_amd.stitch import perform_stitch
    again = perform_stitch(pdir, str(d / "restitch"), "again", 2)
    assert open(again).read().split("\n")[1] == fasta[1]
    # ... and equal to what the ORACLE's labels give when a naive, independent statement of the reference's
    # stitch procedure (tests/naive_stitch.py; alignments from the reference's own SSW when built) joins them
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import naive_stitch
    from helen_amd.data_store import DataStore
    odir = d / "oracle_predictions"
    odir.mkdir()
    with DataStore(str(odir / "oracle_0.hdf"), "w") as store:
        for (cs, chunk), (eb, er, pos) in expected.items():
            store.write_prediction("chr20_synth", cs, cs + 1000, chunk, pos, eb, er)
    want = naive_stitch.stitch_directory(str(odir), threads=1)    # polish ran with -t 1 (helen.py default)
    assert list(want) == ["chr20_synth"] and want["chr20_synth"] == fasta[1]


def test_cli_polish_in_bf16_mode(workdir):
    """BASELINE.json configs[3] through the product's own command: `$HELEN_PRECISION=bf16 helen polish ...` writes the same
    files with the same names and dtypes; its labels are the bf16 ENGINE's labels of the same windows (the command adds
    nothing of its own) and all but a sliver of the fp32 oracle's."""
    from helen_amd.engine import HelenEngine
    from helen_amd.sequence_dataset import SequenceDataset
    d, w, model, img_dir, expected = workdir
    out = str(d / "out_cli_bf16")
    env = dict(os.environ, HELEN_PRECISION="bf16")
    r = subprocess.run([os.path.join(ROOT, "bin", "helen"), "polish", "-i", img_dir, "-m", model, "-b", "32", "-w", "2",
                        "-o", out, "-p", "asm16", "-g", "-d_ids", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    pdir = os.path.join(out, [x for x in os.listdir(out) if x.startswith("predictions_")][0])
    files = [os.path.join(pdir, f) for f in sorted(os.listdir(pdir))]
    assert [os.path.basename(f) for f in files] == ["asm16_0.hdf"]
    ds = SequenceDataset(img_dir)
    items = [ds[i] for i in range(len(ds))]
    eng = HelenEngine(w, device=0, max_windows=128, precision="bf16")
    b16, r16 = eng.polish(torch.from_numpy(np.stack([it[4] for it in items])).cuda())[:2]
    b16, r16 = b16.cpu().numpy(), r16.cpu().numpy()
    eng.close()
    _check_prediction_files(files, {(it[1], it[3]): (b16[i], r16[i], it[5]) for i, it in enumerate(items)})
    same = np.mean([np.mean(b16[i] == expected[(it[1], it[3])][0]) * 0.5 + np.mean(r16[i] == expected[(it[1], it[3])][1]) * 0.5
                    for i, it in enumerate(items)])
    assert same > 0.98, same          # random-init weights: thin margins (the trained-network bar is in test_gpu_parity.py)
    assert os.path.getsize(os.path.join(out, "asm16.fa")) > 1000


def test_drop_in_model_object(workdir):
    """The reference's operator-level call `transducer_model(image_chunk, hidden)`
    (predict_gpu.py:129) on the drop-in object loaded from a reference-format checkpoint."""
    import torch

    import oracle
    from helen_amd.model_handler import ModelHandler
    d, w, model, img_dir, expected = workdir
    m, hs, gl, ep = ModelHandler.load_simple_model(model, 1, 90, 1000, 5, 11)
    m.eval()
    m.to(0)
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, size=(5, 100, 90)).astype(np.float32)
    h = m.init_hidden(5, 1)
    base, rle, h1 = m(torch.from_numpy(x), h)                # CPU tensors are moved, like DDP did
    ob, orl, oh = oracle.gru_chunk_forward(w, x, h.numpy())
    np.testing.assert_allclose(base.cpu().numpy(), ob, atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(rle.cpu().numpy(), orl, atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(h1.cpu().numpy(), oh, atol=1e-4, rtol=0)


def test_two_callers_shard_files_round_robin(workdir):
    """Two callers (both on GPU 0 here; one per GPU in production): files are dealt round-robin
    (CallConsensusInterface.py:138-145), each rank writes <prefix>_<rank>.hdf, the union is complete."""
    from helen_amd.call_consensus import call_consensus
    from helen_amd.file_manager import get_file_paths_from_directory, shard_round_robin
    d, w, model, img_dir, expected = workdir
    out = str(d / "out_two")
    call_consensus(img_dir, model, 16, 0, 1, out, "pred", True, "0,0", 2)
    files = [os.path.join(out, f) for f in sorted(os.listdir(out))]
    assert [os.path.basename(f) for f in files] == ["pred_0.hdf", "pred_1.hdf"]
    _check_prediction_files(files, expected)
    # rank 0 holds exactly the windows of its file shard
    shards = shard_round_robin(get_file_paths_from_directory(img_dir), 2)
    from helen_amd.sequence_dataset import SequenceDataset
    n0 = len(SequenceDataset(None, file_list=shards[0]))
    with hdf5.File(files[0]) as f:
        assert len(f.keys("predictions/chr20_synth")) == n0


def test_evaluation_interface_end_to_end(tmp_path):
    """helen_amd.evaluate.test_interface (TestInterface.py:93-141 / models/test.py) on a labeled image
    directory holding the eval10 golden case: loss, loss sums and both confusion matrices equal what
    torch's CrossEntropyLoss + the reference model gave; the matrices are saved."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_cases import EVAL_BATCH, EVAL_LOSS_RTOL, load_case
    from helen_amd.evaluate import SequenceDataset, test_interface
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_file
    w, img, g = load_case("eval10")
    img_dir = tmp_path / "labeled"
    img_dir.mkdir()
    # two files; names sort in window order so the loader batches are the golden's
    write_image_file(str(img_dir / "a.h5"), img[:6], first_window=100, labels=(g["label_base"][:6], g["label_rle"][:6]))
    write_image_file(str(img_dir / "b.h5"), img[6:], first_window=106, labels=(g["label_base"][6:], g["label_rle"][6:]))
    ds = SequenceDataset(str(img_dir))
    assert len(ds) == 10 and np.array_equal(ds[7][0], img[7]) and np.array_equal(ds[7][2], g["label_rle"][7])
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(w, None, 128, 1, 0, model)
    out = str(tmp_path / "eval_out")
    stats = test_interface(str(img_dir), EVAL_BATCH, True, 0, model, out, False)
    np.testing.assert_allclose(stats["loss"], g["loss"][0], rtol=EVAL_LOSS_RTOL)
    np.testing.assert_allclose(stats["total_loss_rle"], g["total_loss_rle"][0], rtol=EVAL_LOSS_RTOL)
    assert stats["total_images"] == int(g["total_images"][0]) and stats["accuracy"] == 0
    assert np.array_equal(stats["base_confusion_matrix"], g["base_confusion_matrix"])
    assert np.array_equal(stats["rle_confusion_matrix"], g["rle_confusion_matrix"])
    saved = np.loadtxt(os.path.join(out, "RLE_CONFUSION_MATRIX.tsv"), dtype=np.int64)
    assert np.array_equal(saved, g["rle_confusion_matrix"])


def _predict_fixture():
    import gzip
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_golden_predict",
                                                  os.path.join(ROOT, "tests", "golden", "make_golden_predict.py"))
    gen = importlib.util.module_from_spec(spec)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    spec.loader.exec_module(gen)          # only its input builder and tree reader are used: nothing of the reference
    with gzip.open(os.path.join(ROOT, "tests", "golden", "predict_ref.json.gz"), "rt") as f:
        return gen, json.load(f)


TIE_MARGIN = 2e-6      # the stated bar of tests/test_gpu_scale.py: a label may differ only on a tie of the accumulators


def _assert_same_tree(got, want, what, margins=None):
    """Names, dtypes, shapes and bytes of every dataset; a label dataset may differ from the reference's only at
    positions where `margins(path)` (the oracle's top-1 / top-2 margin of the accumulated softmax there) is below
    TIE_MARGIN.  -> number of such tie positions"""
    import base64
    assert sorted(got) == sorted(want), (what, sorted(set(got) ^ set(want))[:6])
    ties = 0
    for path in sorted(want):
        g, w = got[path], want[path]
        if g["sha1"] != w["sha1"] and "b64" in w:
            a = np.frombuffer(base64.b64decode(g["b64"]), np.uint8)
            b = np.frombuffer(base64.b64decode(w["b64"]), np.uint8)
            where = np.flatnonzero(a != b)
            m = margins(path)[where] if margins is not None else None
            if m is None or float(m.max()) >= TIE_MARGIN:
                raise AssertionError("%s: %s: %d of %d labels differ from the reference's, at %s, margins %s"
                                     % (what, path, len(where), a.size, where[:5], None if m is None else m[:5]))
            print("%s: %s: position(s) %s differ from the reference's on a tie (margin %s)" % (what, path, where, m))
            ties += len(where)
            continue
        assert (g["dtype"], g["shape"], g["sha1"]) == (w["dtype"], w["shape"], w["sha1"]), (what, path)
    return ties


@pytest.mark.parametrize("workers", [0, 2])
def test_prediction_file_equals_the_reference_predict_itself(tmp_path, workers):
    """tests/golden/predict_ref.json.gz is the prediction file the REFERENCE's own `predict` (models/predict.py:38-175: its
    reader, DataLoader batches of 4, model loader, 19-chunk loop, softmax / zero-pad-add / argmax, its writer) wrote on
    CPU for the image directory and the `.pkl` of make_golden_predict.predict_case.  `helen_amd.predict.predict` on the
    MI355X, from the same directory and model file, must write the same tree: region and chunk names, scalar bounds,
    uint32 positions with wrapped padding, and the labels byte for byte -- except on exact ties of the accumulated softmax (the
    stated bar of DESIGN.md 5: the one differing label of 44,000 sits where the reference's own margin is 3.4e-7)."""
    from helen_amd.predict import predict
    gen, fixture = _predict_fixture()
    image_dir, model = gen.predict_case(str(tmp_path))
    files = sorted(os.path.join(image_dir, f) for f in os.listdir(image_dir))
    out = str(tmp_path / "hip")
    predict(files, out, model, fixture["batch"], workers, 0, 0)

    def margins(path):
        # the oracle's accumulators of that window (it reproduces the reference's file exactly:
        # tests/test_predict_pipeline_cpu.py::test_oracle_reader_and_writer_reproduce_the_reference_predict_itself)
        import oracle
        from helen_amd.sequence_dataset import SequenceDataset, _load_batch
        from helen_amd.weights import make_weights
        _, _, contig, region, chunk, kind = path.split("/")
        ds = SequenceDataset(image_dir)
        k = [n for _, n in ds.all_images].index(region + "-" + chunk)
        b = _load_batch(ds.all_images[k:k + 1])
        ref = oracle.polish_batch(make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0), b.images)
        acc = np.sort(ref["acc_base" if kind == "bases" else "acc_rle"][0], axis=-1)
        return acc[:, -1] - acc[:, -2]
    ties = _assert_same_tree(gen.tree_of(out + "_0.hdf"), fixture["tree"], "helen_amd.predict on the GPU", margins)
    assert ties <= 2          # of 44,000 labels (measured: 1, margin 3.4e-7)


def test_prediction_file_equals_the_reference_predict_at_scale(tmp_path):
    """tests/golden/predict_ref_large.npz holds the labels the REFERENCE's own `predict` (models/predict.py:38-175, run on
    CPU in the build container by make_golden_predict_large.py) wrote for 4,096 seeded windows = 8.19 M labels.
    `helen_amd.predict.predict` on the MI355X, from the same image directory (regenerated here from the seeds) and the
    same `.pkl`, must write the same tree (names, bounds, uint32 positions: one digest) and the same labels.  Where a
    label differs, the float64 evaluation of the network (oracle_polish_batch_f64) arbitrates: the stated bar is that
    the HIP path is wrong only where the float64 top-1 / top-2 margin is below fp32 resolution (1e-6), that it is wrong
    no more often than the reference's own fp32 arithmetic, and that at most 2e-6 of the labels differ.  The table (who float64 sides with) is printed."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import make_golden_predict_large as G
    from golden_cases import arbitrate_label_differences, assert_wrong_only_below_fp32_resolution
    from helen_amd.predict import predict
    fx = np.load(os.path.join(ROOT, "tests", "golden", "predict_ref_large.npz"))
    image_dir, model, images = G.large_case(str(tmp_path))
    files = sorted(os.path.join(image_dir, f) for f in os.listdir(image_dir))
    out = str(tmp_path / "hip")
    predict(files, out, model, int(fx["batch"]), 4, 0, 0)
    bases, rles, rest = G.labels_of(out + "_0.hdf")
    assert rest == str(fx["rest_sha1"]), "region names / bounds / positions differ from the reference's file"
    rows, summary = arbitrate_label_differences(
        G.large_weights(), images, {"bases": bases, "rles": rles}, {"bases": fx["bases"], "rles": fx["rles"]},
        "hip", "reference")
    # measured: 2 of 8,192,000 differ (2.4e-7), float64 sides with the HIP path both times (margins 6.4e-8, 1.2e-7)
    assert summary["rate"] <= 2e-6, summary
    hip_wrong = assert_wrong_only_below_fp32_resolution(rows, "a")
    assert hip_wrong <= max(3, 2 * (len(rows) - hip_wrong)), summary      # not worse than the reference's own fp32


def test_two_rank_end_to_end(tmp_path):
    """The multi-rank product path at a size where the pipeline is in steady state: call_consensus over TWO callers
    (both on GPU 0 here, one per GPU in production) with reader processes, 2 x 6,144 windows.  The union of
    `<prefix>_0.hdf` and `<prefix>_1.hdf` holds every window once with the labels a direct engine call gives for the
    same images (and, on a 192-window sample, the oracle's); the host plan and both ranks' stage seconds are
    reported (helen_amd.predict.LAST_RUN)."""
    import torch

    import oracle
    from helen_amd import predict as P
    from helen_amd.call_consensus import call_consensus
    from helen_amd.engine import HelenEngine
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_dir
    from helen_amd.weights import make_images
    w = make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0)
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(w, None, 128, 1, 0, model)
    per_file, n_files = 1536, 8
    img_dir = str(tmp_path / "img")
    write_image_dir(img_dir, per_file * n_files, n_files=n_files, seed=900, direct=True)
    out = str(tmp_path / "out")
    call_consensus(img_dir, model, 256, 3, 1, out, "p", True, "0,0", 2)
    run = dict(P.LAST_RUN)
    assert [r["rank"] for r in run["ranks"]] == [0, 1]
    assert [r["windows"] for r in run["ranks"]] == [per_file * n_files // 2] * 2
    assert run["host_plan"]["n_ranks"] == 2 and all(k >= 1 for k in run["host_plan"]["reader_workers_per_rank"])
    print("two-rank run: %.2f s; plan %s; ranks %s" % (run["seconds"], run["host_plan"], run["ranks"]))
    files = sorted(os.listdir(out))
    assert files == ["p_0.hdf", "p_1.hdf"]
    eng = HelenEngine(w, device=0, max_windows=4096)
    seen = 0
    for fi in range(n_files):
        img = make_images(per_file, seed=900 + fi)
        b, r = eng.polish(torch.from_numpy(img).cuda())
        b, r = b.cpu().numpy(), r.cpu().numpy()
        ref = oracle.polish_batch(w, img[:24]) if fi < 2 else None
        with hdf5.File(os.path.join(out, files[fi % 2])) as f:      # file fi goes to rank fi % 2
            for i in range(per_file):
                start = 800 * (fi * per_file + i)
                root = "predictions/chr20_synth/chr20_synth-%d-%d/0/" % (start, start + 1000)
                gb, gr = f.read(root + "bases"), f.read(root + "rles")
                assert np.array_equal(gb, b[i]) and np.array_equal(gr, r[i]), (fi, i)
                if ref is not None and i < 24:
                    assert np.array_equal(gb, ref["bases"][i]) and np.array_equal(gr, ref["rles"][i]), (fi, i)
                seen += 1
    eng.close()
    total = 0
    for name in files:
        with hdf5.File(os.path.join(out, name)) as f:
            total += len(f.keys("predictions/chr20_synth"))
    assert seen == total == per_file * n_files


def test_bench_two_ranks_end_to_end_line():
    """`bench.py --gpus 2 --single-device --e2e N`: the JSON line carries the N-rank end-to-end leg."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-host-path", "--e2e", "8192", "--e2e-workers", "3"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    e = line["end_to_end"]
    # (the leg's input is a simulated assembly of three-image regions: about the windows asked for, in whole regions)
    assert line["n_gpus"] == 2 and e["n_ranks"] == 2 and 0.97 * 16384 <= e["windows"] <= 16384, e
    assert e["regions_stored"] * 3 == e["windows"] and e["weights"] == "trained_synth"
    assert e["output_files"] == ["p_0.hdf", "p_1.hdf"] and len(e["per_rank"]) == 2
    # the whole `polish` of the same directory, stitch pipelined behind the inference: the FASTA of the two-phase stitch
    assert e["polish"]["fasta_equals_two_phase"] is True and e["polish_seconds"] > 0
    assert all(r_["stitch_stream"]["regions"] * 3 == r_["windows"] for r_ in e["polish"]["per_rank"]), e["polish"]
    # two ranks: their regions went to stitch collector processes while they ran (helen_amd/stitch_collect.py)
    sc = e["polish"]["stitch_collectors"]
    assert sc["collectors"] >= 1 and sum(c["regions"] for c in sc["per_collector"]) * 3 == e["windows"], sc
    assert all(c["aligned_now"] == 0 for c in sc["per_collector"]), sc       # every join was aligned behind the inference
    assert e["value"] > 0 and e["usable_cpus"] >= 1 and e["predicted_host_ceiling"] > 0
    assert line["barrier"] == "gloo all-reduce" and line["rccl"].startswith("not used"), line["rccl"]
    print(json.dumps(e))


def test_bench_eight_ranks_end_to_end_line():
    """BASELINE.json configs[2] without the node: `bench.py --gpus 8 --single-device --e2e 8192` -- eight bench ranks,
    then the product's call_consensus over eight spawned ranks (all on cuda:0), each with the readers the host plan grants,
    each writing its own prediction file; the line carries all eight ranks and every window."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--single-device", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--no-host-path", "--no-margins", "--e2e", "8192",
                        "--e2e-workers", "8"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    e = line["end_to_end"]
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and len(line["per_rank_windows_per_s"]) == 8
    assert e["n_ranks"] == 8 and 0.97 * 8 * 8192 <= e["windows"] <= 8 * 8192 and e["regions_stored"] * 3 == e["windows"], e
    assert e["output_files"] == ["p_%d.hdf" % k for k in range(8)] and len(e["per_rank"]) == 8
    assert all(r_["windows"] * 8 == e["windows"] and r_["reader_workers"] >= 1 for r_ in e["per_rank"])
    assert e["polish"]["fasta_equals_two_phase"] is True
    assert sum(c["regions"] for c in e["polish"]["stitch_collectors"]["per_collector"]) * 3 == e["windows"]
    assert sum(e["reader_workers_per_rank"]) + 2 * 8 <= max(e["usable_cpus"], 3 * 8)       # the host budget holds
    assert e["predicted_bound"] in ("device", "host readers")
    print(json.dumps({k: e[k] for k in ("value", "usable_cpus", "reader_workers_per_rank", "predicted_bound")}))


def test_bench_survives_an_rccl_that_does_not_come_up():
    """The barrier of `bench.py --gpus N` crosses RCCL AND gloo; the times travel over gloo.  Two ranks on ONE device
    is a configuration RCCL refuses: the line must still come out, say so, and have used the gloo barrier (on a node
    whose RCCL works the same code prints "rccl": "ok")."""
    import json
    env = dict(os.environ, HELEN_BENCH_TRY_RCCL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-host-path", "--no-margins", "--e2e", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value"] > 0
    assert line["rccl"] == "ok" or line["rccl"].startswith("failed"), line["rccl"]
    assert line["barrier"] == ("RCCL all-reduce + gloo all-reduce" if line["rccl"] == "ok" else "gloo all-reduce")
    print(line["rccl"], line["barrier"])


def test_slot_pipeline_equals_device_calls():
    """helen_polish_slot_submit / _wait (the hand-off `helen polish` runs on, helen_amd/native_engine.py): slots of 4096, 1000
    and 17 windows in page-locked memory of helen_host_alloc, two in flight, give the labels of helen_polish_batch on the
    same windows; a third submission without a wait, pageable buffers, a wait with nothing in flight and a
    helen_polish_host while slots are in flight are refused -- and the handle works on afterwards."""
    from helen_amd._lib import HelenError
    from helen_amd.engine import HelenEngine
    from helen_amd.native_engine import NativeEngine, PinnedBlock, device_count
    from helen_amd.weights import make_images
    assert device_count() >= 1
    w = make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0)
    sizes = [4096, 1000, 17, 4096]
    img = make_images(max(sizes), seed=77)
    ref_eng = HelenEngine(w, device=0, max_windows=4096)
    b, r = ref_eng.polish(torch.from_numpy(img).cuda())
    want_b, want_r = b.cpu().numpy(), r.cpu().numpy()
    ref_eng.close()
    eng = NativeEngine(w, device=0, max_windows=4096)
    blocks, slots = [], []
    for n in sizes:
        blk = PinnedBlock(n * 92000 + 4096)
        blocks.append(blk)
        images = blk.array[:n * 90000].reshape(n, 1000, 90)
        images[:] = img[:n]
        bases = blk.array[n * 90000:n * 91000].reshape(n, 1000)
        rles = blk.array[n * 91000:n * 92000].reshape(n, 1000)
        bases[:] = 255
        rles[:] = 255
        slots.append((images, bases, rles))
    eng.submit(*slots[0])
    eng.submit(*slots[1])
    with pytest.raises(HelenError, match="two slots are in flight"):
        eng.submit(*slots[2])
    with pytest.raises(HelenError, match="slots are in flight"):
        eng.polish_host(img[:8])
    eng.wait()
    eng.submit(*slots[2])
    eng.wait()
    eng.submit(*slots[3])
    eng.wait()
    eng.wait()
    assert eng.in_flight == 0
    with pytest.raises(HelenError, match="no slot is in flight"):
        eng.wait()
    for n, (images, bases, rles) in zip(sizes, slots):
        assert np.array_equal(bases, want_b[:n]) and np.array_equal(rles, want_r[:n]), n
    pageable = np.ascontiguousarray(img[:32])
    with pytest.raises(HelenError, match="page-locked"):
        eng.submit(pageable, np.empty((32, 1000), np.uint8), np.empty((32, 1000), np.uint8))
    assert eng.in_flight == 0
    got_b, got_r = eng.polish_host(pageable)                      # the handle is fine afterwards
    assert np.array_equal(got_b, want_b[:32]) and np.array_equal(got_r, want_r[:32])
    eng.close()
    for blk in blocks:
        blk.close()


def test_predict_on_the_native_stage_equals_the_torch_stage(tmp_path, monkeypatch):
    """helen_amd.predict.predict on the library's slot pipeline (the default: no torch in the data path) writes the prediction
    file of round 4's torch device stage ($HELEN_DEVICE_STAGE=torch), byte for byte, on a directory with short images, several
    device calls and a ragged last one."""
    import hashlib

    from helen_amd import predict as P
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_dir
    w = make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0)
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(w, None, 128, 1, 0, model)
    img_dir = str(tmp_path / "img")
    files = write_image_dir(img_dir, 9000, n_files=3, seed=5, short_every=9, direct=True)
    digests = {}
    for stage in ("native", "torch"):
        monkeypatch.setenv("HELEN_DEVICE_STAGE", stage)
        out = str(tmp_path / stage)
        P.predict(files, out, model, 256, 3, 0, 0)
        assert P.LAST_PREDICT["windows"] == 9000 and P.LAST_PREDICT["device_calls"] == 3
        digests[stage] = hashlib.sha1(open(out + "_0.hdf", "rb").read()).hexdigest()
    assert digests["native"] == digests["torch"]
