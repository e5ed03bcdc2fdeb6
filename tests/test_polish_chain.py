"""`helen polish` end to end -- image directory -> prediction HDF5 -> FASTA -- against the REFERENCE's own chain.

tests/golden/polish_ref.json.gz (made in the build container by tests/golden/make_golden_polish.py) holds what the
reference's `models/predict.py` followed by its `StitchInterface.perform_stitch` (the chain of PolishInterface.py:49-105)
produced for the simulated assembly helen_amd.synthetic.POLISH_CASE on the trained network of
tests/golden/trained_synth.npz: the prediction tree, its labels, the FASTA.  The tests here run the product's command
line on the same image directory (regenerated from the seeds) and model file: same tree, same labels, FASTA byte for
byte.  On the host engine (no GPU) here; on the MI355X, in bf16 and over two callers under `-m gpu`.
"""
import base64
import gzip
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def _fixture():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_polish", os.path.join(ROOT, "tests", "golden", "make_golden_polish.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)          # only its input builder and tree reader are used: nothing of the reference
    with gzip.open(os.path.join(ROOT, "tests", "golden", "polish_ref.json.gz"), "rt") as f:
        return gen, json.load(f)


def _helen(args, env=None, expect=0):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "helen")] + args, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == expect, p.stderr[-3000:]
    return p


def _prediction_files(out_dir):
    pred = [d for d in os.listdir(out_dir) if d.startswith("predictions_")]
    assert len(pred) == 1, pred
    pred = os.path.join(out_dir, pred[0])
    return sorted(os.path.join(pred, f) for f in os.listdir(pred) if f.endswith(".hdf"))


def _merged_tree(gen, files):
    """The union of the trees of one or several prediction files (every dataset exactly once)."""
    tree, paths, labels = {}, [], {}
    for path in files:
        t, lp, lab = gen.tree_and_labels(path)
        assert not set(t) & set(tree), "a dataset is in two prediction files"
        tree.update(t)
        at = 0
        for p in lp:
            n = int(np.prod(t[p]["shape"]))
            labels[p] = lab[at:at + n]
            at += n
    return tree, labels


def _label_differences(fixture, labels):
    want = zlib.decompress(base64.b64decode(fixture["labels_zb64"]))
    diff, at = [], 0
    for p in fixture["label_paths"]:
        n = len(labels[p])
        a, b = np.frombuffer(labels[p], np.uint8), np.frombuffer(want[at:at + n], np.uint8)
        at += n
        for k in np.flatnonzero(a != b):
            diff.append((p, int(k), int(a[k]), int(b[k])))
    return diff


def _assert_chain_equals_reference(gen, fixture, out_dir, threads, what):
    tree, labels = _merged_tree(gen, _prediction_files(out_dir))
    assert sorted(tree) == sorted(fixture["tree"]), (what, sorted(set(tree) ^ set(fixture["tree"]))[:6])
    diff = _label_differences(fixture, labels)
    assert not diff, "%s: %d labels differ from the reference's predict(): %s" % (what, len(diff), diff[:8])
    for p, w in fixture["tree"].items():
        g = tree[p]
        assert (g["dtype"], g["shape"], g["sha1"]) == (w["dtype"], w["shape"], w["sha1"]), (what, p)
    fasta = gen.read_fasta(os.path.join(out_dir, "polished.fa"))
    assert fasta == fixture["fasta"][str(threads)], "%s: the FASTA differs from the reference chain's" % what
    return fasta


def _identity_report(fasta, truth, what):
    from edit_distance import banded_edit_distance
    seqs, name = {}, None
    for line in fasta.splitlines():
        if line.startswith(">"):
            name = line[1:]
        else:
            seqs[name] = line
    total = errors = 0
    for contig, t in truth.items():
        if contig == "ctgA":                # a hole of ~3,700 bases by construction: lengths only
            print("%s: %s: truth %d bases, polished %d (one region is left out)" % (what, contig, len(t), len(seqs[contig])))
            continue
        d = banded_edit_distance(t, seqs[contig], band=512)
        print("%s: %s: truth %d bases, polished %d, edit distance %s" % (what, contig, len(t), len(seqs[contig]), d))
        total += len(t)
        errors += d
    print("%s: identity to the simulated truth outside the hole: %.4f" % (what, 1.0 - errors / total))
    return 1.0 - errors / total


def test_assembly_generator_is_deterministic_and_writers_agree(tmp_path):
    """The simulated assembly: both file writers (libhdf5, direct emitter) give the reader the same windows; every
    region's images share a file; the rows cover the contig; the row keys are in stitch's order."""
    from helen_amd import synthetic as S
    from helen_amd.sequence_dataset import SequenceDataset
    a = S.write_assembly_dir(str(tmp_path / "lib"), S.POLISH_CASE, S.POLISH_CASE_FILES, blocks=S.POLISH_CASE_BLOCKS)
    b = S.write_assembly_dir(str(tmp_path / "direct"), S.POLISH_CASE, S.POLISH_CASE_FILES, blocks=S.POLISH_CASE_BLOCKS, direct=True)
    assert a["windows"] == b["windows"] == 103 and a["truth"] == b["truth"]
    da, db = SequenceDataset(str(tmp_path / "lib")), SequenceDataset(str(tmp_path / "direct"))
    assert len(da) == len(db) == a["windows"]
    file_of_region, chunk_ids, short, inserts, splits = {}, {}, 0, 0, 0
    for i in range(len(da)):
        x, y = da[i], db[i]
        assert x[:4] == y[:4]
        assert np.array_equal(np.asarray(x[4]), np.asarray(y[4])) and np.array_equal(np.asarray(x[5]), np.asarray(y[5]))
        key = x[:3]
        assert file_of_region.setdefault(key, os.path.basename(x[6])) == os.path.basename(x[6])
        chunk_ids.setdefault(key, []).append(x[3])
        pos = np.asarray(x[5])
        live = pos[:, 0] >= 0
        short += int(not live.all())
        inserts += int((pos[live][:, 1] > 0).sum())
        splits += int((pos[live][:, 2] > 0).sum())
        keys = [tuple(r) for r in pos[live].tolist()]
        assert keys == sorted(keys)
    assert max(len(v) for v in chunk_ids.values()) == 13 and short > 30 and inserts > 5000 and splits > 20
    assert len(set(file_of_region.values())) == 3
    c = S.assembly_contigs(S.POLISH_CASE)[0]
    assert len(c.regions) == 13 and c.truth() == a["truth"]["ctgA"]        # 14 regions, one left out


@pytest.mark.parametrize("threads,callers", [(3, 1), (3, 2)])
def test_host_polish_equals_the_reference_chain(tmp_path, threads, callers):
    """`helen polish` WITHOUT --gpu_mode (the product's host engine, libhelen_cpu.so) on the simulated assembly: the
    prediction file and the FASTA of the reference's own predict + perform_stitch, byte for byte.  With two callers
    (two processes, two prediction files) the regions travel to a stitch collector process while the callers run
    (helen_amd/stitch_collect.py) -- the path of a multi-GPU `polish`."""
    gen, fixture = _fixture()
    image_dir, model, made = gen.polish_case(str(tmp_path))
    out = str(tmp_path / "out")
    r = _helen(["polish", "-i", image_dir, "-m", model, "-b", "16", "-w", "0", "-t", str(threads), "-c", str(callers), "-o", out,
                "-p", "polished"], env={"HELEN_ASSERT_NO_TORCH": "1"})
    assert len(_prediction_files(out)) == callers
    if callers > 1 and r is not None:
        assert "STITCH COLLECTOR(S) OVER 2 RANK(S)" in r.stderr, r.stderr[-600:]
    fasta = _assert_chain_equals_reference(gen, fixture, out, threads, "host path")
    assert _identity_report(fasta, made["truth"], "host path") > 0.995


@pytest.mark.parametrize("callers", [1, 2])
def test_a_failing_stitch_stage_does_not_fail_polish(tmp_path, callers):
    """The stitch stage behind the inference is an optimisation: when it stops (a full spill directory -- injected here
    at the second device call of every caller) the prediction files are still written whole, the command says so, stitches
    them in a second phase and exits 0 with the same FASTA."""
    gen, fixture = _fixture()
    image_dir, model, made = gen.polish_case(str(tmp_path))
    out = str(tmp_path / "out")
    r = _helen(["polish", "-i", image_dir, "-m", model, "-b", "16", "-w", "0", "-t", "3", "-c", str(callers), "-o", out,
                "-p", "polished"], env={"HELEN_DEBUG_HOOKS": "1", "HELEN_DEBUG_STITCH_FAIL": "1"})
    assert "THE STITCH STAGE BEHIND THE INFERENCE STOPPED" in r.stderr, r.stderr[-1500:]
    assert "No space left on device (injected" in r.stderr
    assert len(_prediction_files(out)) == callers
    _assert_chain_equals_reference(gen, fixture, out, 3, "host path, stitch stage stopped")


@pytest.mark.gpu
def test_gpu_polish_equals_the_reference_chain(tmp_path):
    """`helen polish -g` on the MI355X, the command a user types: same prediction tree and labels as the reference's
    predict(), FASTA byte-identical to the reference chain's; then the same over two callers sharing the device (files
    sharded round-robin, two prediction files), and from image files written by the direct emitter."""
    gen, fixture = _fixture()
    image_dir, model, made = gen.polish_case(str(tmp_path))
    for what, extra in (("one caller", []), ("two callers on one device", ["-d_ids", "0,0"])):
        out = str(tmp_path / ("out_" + what.split()[0]))
        # (HELEN_ASSERT_NO_TORCH: the command must get through without importing torch -- checkpoint, slots and device stage
        # are the library's own, helen_amd/native_engine.py; exit code 3 otherwise)
        _helen(["polish", "-i", image_dir, "-m", model, "-b", "256", "-w", "2", "-t", "3", "-o", out, "-p", "polished", "-g"] + extra,
               env={"HELEN_ASSERT_NO_TORCH": "1"})
        assert len(_prediction_files(out)) == (2 if extra else 1)
        fasta = _assert_chain_equals_reference(gen, fixture, out, 3, "MI355X, " + what)
    assert _identity_report(fasta, made["truth"], "MI355X fp32") > 0.995


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp32x3"])
def test_gpu_polish_reduced_precision_against_the_reference_chain(tmp_path, precision):
    """The same command with --precision bf16 / fp32x3: labels that differ from the reference's fp32 predict() are
    counted and the FASTA's distance from the reference chain's is printed (BASELINE.json configs[3]: argmax parity is a
    reported figure there, not an identity)."""
    from edit_distance import banded_edit_distance
    gen, fixture = _fixture()
    image_dir, model, made = gen.polish_case(str(tmp_path))
    out = str(tmp_path / "out")
    _helen(["polish", "-i", image_dir, "-m", model, "-b", "512", "-w", "2", "-t", "3", "-o", out, "-p", "polished", "-g",
            "--precision", precision])
    tree, labels = _merged_tree(gen, _prediction_files(out))
    assert sorted(tree) == sorted(fixture["tree"])
    diff = _label_differences(fixture, labels)
    n = sum(len(v) for v in labels.values())
    fasta = gen.read_fasta(os.path.join(out, "polished.fa"))
    ref = fixture["fasta"]["3"]
    d = banded_edit_distance(ref.replace("\n", "|"), fasta.replace("\n", "|"), band=512)
    print("%s: %d of %d labels differ from the reference's fp32 predict(); FASTA edit distance to the reference chain's: %s "
          "of %d bytes" % (precision, len(diff), n, d, len(ref)))
    assert len(diff) <= (2 if precision == "fp32x3" else n * 2e-3)
    assert d is not None and d <= (4 if precision == "fp32x3" else 400)
    _identity_report(fasta, made["truth"], precision)
