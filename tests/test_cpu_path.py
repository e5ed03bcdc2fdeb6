"""The host path of runs WITHOUT --gpu_mode (libhelen_cpu.so: helen_amd/csrc/cpu_path.cpp behind include/helen_cpu.h; the
reference's counterpart is the ONNX Runtime session of models/predict_cpu.py:39-170) against the golden vectors of the
REFERENCE's own TransducerGRU (tests/golden/*.npz) -- the fixtures the oracle and the HIP path are held to, the same
stated fp32 tolerance, labels identical -- and BASELINE.json configs[0] end to end through the `helen` command: 100
windows at batch 4, no GPU.  The product's own code: nothing here or there touches oracle/."""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from golden_cases import ACC_ATOL, HIDDEN_ATOL, LOGIT_ATOL, LOGIT_RTOL, label_mismatch_report, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(w, threads=4):
    from helen_amd.cpu_engine import CpuEngine
    return CpuEngine(w, threads=threads)


def test_cpu_library_exports_what_its_header_declares():
    from helen_amd import cpu_engine
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "helen_cpu.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(helen_cpu_[a-z_0-9]+)\s*\(", header))
    assert declared == set(cpu_engine.EXPORTS)
    lib = cpu_engine.load()
    for name in declared:
        assert hasattr(lib, name), name
    src = open(os.path.join(ROOT, "helen_amd", "csrc", "cpu_path.cpp")).read() + open(os.path.join(ROOT, "helen_amd", "cpu_engine.py")).read()
    assert not re.search(r"^\s*(#include|import|from)\b.*oracle", src, flags=re.M)     # the product's own code


@pytest.mark.parametrize("case", ["trace6", "small_input6"])
def test_host_path_matches_reference_traces(case):
    """The reference's 19-chunk loop driven through the operator entry (hidden carried chunk to chunk, predict_cpu.py:114-118)
    and through the whole-batch entry: hidden after every chunk, logits of chunks 0 / 9 / 18, accumulated softmax, labels."""
    w, img, g = load_case(case)
    e = _engine(w)
    x = img.astype(np.float32)
    h = np.zeros((img.shape[0], 2, 128), np.float32)
    hidden, lb, lr = [], {}, {}
    for c in range(19):
        base, rle, h = e.chunk_forward(x[:, 50 * c:50 * c + 100], h)
        hidden.append(h.copy())
        if c in (0, 9, 18):
            lb[c], lr[c] = base, rle
    np.testing.assert_allclose(np.stack(hidden), g["hidden"], atol=HIDDEN_ATOL, rtol=0)
    np.testing.assert_allclose(np.stack([lb[c] for c in (0, 9, 18)]), g["logit_base"], atol=LOGIT_ATOL, rtol=LOGIT_RTOL)
    np.testing.assert_allclose(np.stack([lr[c] for c in (0, 9, 18)]), g["logit_rle"], atol=LOGIT_ATOL, rtol=LOGIT_RTOL)
    bases, rles, ab, ar = e.polish_host(img, want_acc=True)
    np.testing.assert_allclose(ab[:3], g["acc_base"], atol=ACC_ATOL, rtol=0)
    np.testing.assert_allclose(ar[:3], g["acc_rle"], atol=ACC_ATOL, rtol=0)
    nb, rep = label_mismatch_report(g["acc_base"], g["bases"], bases, "base")
    assert nb == 0, rep
    nr, rep = label_mismatch_report(g["acc_rle"], g["rles"], rles, "rle")
    assert nr == 0, rep
    # one forward with T = 37 and a non-zero incoming hidden
    base, rle, h = e.chunk_forward(g["fwd_x"], g["fwd_h0"])
    np.testing.assert_allclose(base, g["fwd_base"], atol=LOGIT_ATOL, rtol=LOGIT_RTOL)
    np.testing.assert_allclose(rle, g["fwd_rle"], atol=LOGIT_ATOL, rtol=LOGIT_RTOL)
    np.testing.assert_allclose(h, g["fwd_h"], atol=HIDDEN_ATOL, rtol=0)


def test_host_path_on_a_trained_network_and_ragged_batches():
    from golden_cases import load_trained_synth
    w, g = load_trained_synth()
    e = _engine(w, threads=3)
    bases, rles, ab, ar = e.polish_host(g["images"], want_acc=True)          # 8 windows: one ragged block
    assert np.array_equal(bases, g["bases"]) and np.array_equal(rles, g["rles"])
    np.testing.assert_allclose(ab, g["acc_base"], atol=ACC_ATOL, rtol=0)
    np.testing.assert_allclose(ar, g["acc_rle"], atol=ACC_ATOL, rtol=0)
    # windows never interact, whatever the block they land in and however many threads run: 1, 17, 33 windows
    w2, img, _ = load_case("config1_100")
    e2 = _engine(w2, threads=2)
    full = e2.polish_host(img[:33], want_acc=True)
    for n in (1, 17):
        part = _engine(w2, threads=5).polish_host(img[:n], want_acc=True)
        for a, b in zip(full, part):
            assert np.array_equal(a[:n], b)
    with pytest.raises(RuntimeError, match="TRAIN_WINDOW"):
        e2.chunk_forward(np.zeros((1, 101, 90), np.float32), np.zeros((1, 2, 128), np.float32))


def test_config0_through_the_helen_command_without_a_gpu(tmp_path):
    """BASELINE.json configs[0]: a synthetic 100-window image file, batch 4, the CPU path -- `helen polish` (no -g) from
    the image directory to the prediction HDF5 and the FASTA, two callers of two threads; labels = the reference's
    (tests/golden/config1_100.npz, its loop run at batch 4)."""
    from helen_amd import hdf5
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_file
    w, img, g = load_case("config1_100")
    img_dir = tmp_path / "img"
    img_dir.mkdir()
    write_image_file(str(img_dir / "a.h5"), img[:60], first_window=0)
    write_image_file(str(img_dir / "b.h5"), img[60:], first_window=60)
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(w, None, 128, 1, 0, model)
    out = tmp_path / "out"
    r = subprocess.run([os.path.join(ROOT, "bin", "helen"), "polish", "-i", str(img_dir), "-m", model, "-b", "4", "-w", "0",
                        "-t", "4", "-c", "2", "-o", str(out), "-p", "asm"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "HOST PATH: 2 CALLER(S) OF 2 THREAD(S)" in r.stderr
    preds = sorted(glob.glob(str(out / "predictions_*" / "asm_*.hdf")))
    assert [os.path.basename(p) for p in preds] == ["asm_0.hdf", "asm_1.hdf"]
    seen = 0
    for p in preds:
        with hdf5.File(p) as f:
            for region in f.keys("predictions/chr20_synth"):
                k = int(region.split("-")[1]) // 800
                assert np.array_equal(f.read("predictions/chr20_synth/%s/0/bases" % region), g["bases"][k])
                assert np.array_equal(f.read("predictions/chr20_synth/%s/0/rles" % region), g["rles"][k])
                seen += 1
    assert seen == 100
    fasta = open(str(out / "asm.fa")).read()
    assert fasta.startswith(">chr20_synth\n") and len(fasta) > 1000
    # --gpu_mode without a GPU is an error, never a switch to this path
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(ROOT, "bin", "helen"), "call_consensus", "-i", str(img_dir), "-m", model, "-g",
                            "-o", str(tmp_path / "o2")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "NO MI355X VISIBLE" in r.stderr
        assert not glob.glob(str(tmp_path / "o2" / "*.hdf"))


def test_evaluation_on_the_host_matches_the_reference_golden(tmp_path):
    """`helen_train test` WITHOUT --gpu_mode: the loop of models/test.py:78-126 on the host engine against
    tests/golden/eval10.npz (the reference model with torch's own CrossEntropyLoss, loader batch 4 -> batches 4 / 4 / 2):
    per-(batch, chunk) losses, loss sums, both confusion matrices -- the bar the device path's evaluation is held to --
    and the whole command through its entry point on a labeled image directory."""
    from golden_cases import EVAL_BATCH, EVAL_LOSS_RTOL
    from helen_amd.evaluate import host_batch_losses
    from helen_amd.options import TrainOptions
    w, img, g = load_case("eval10")
    e = _engine(w, threads=4)
    cm_b, cm_r = np.zeros((5, 5), np.int64), np.zeros((11, 11), np.int64)
    lb, lr = [], []
    for lo in range(0, 10, EVAL_BATCH):
        a, b = host_batch_losses(e, img[lo:lo + EVAL_BATCH], g["label_base"][lo:lo + EVAL_BATCH],
                                 g["label_rle"][lo:lo + EVAL_BATCH], TrainOptions.CLASS_WEIGHTS, cm_b, cm_r)
        lb.append(a)
        lr.append(b)
    loss_b, loss_r = np.array(lb), np.array(lr)
    np.testing.assert_allclose(np.stack([loss_b.ravel(), loss_r.ravel()], axis=1), g["chunk_losses"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(float((loss_b + loss_r).sum()), g["total_loss"][0], rtol=EVAL_LOSS_RTOL)
    np.testing.assert_allclose(float(loss_r.sum()), g["total_loss_rle"][0], rtol=EVAL_LOSS_RTOL)
    assert np.array_equal(cm_b, g["base_confusion_matrix"]) and np.array_equal(cm_r, g["rle_confusion_matrix"])
    # the command: labeled image directory -> loss and confusion matrices (.tsv)
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_file
    d = tmp_path / "labeled"
    d.mkdir()
    write_image_file(str(d / "a.h5"), img, labels=(g["label_base"], g["label_rle"]))
    model = str(tmp_path / "m.pkl")
    ModelHandler.save_model(w, None, 128, 1, 0, model)
    r = subprocess.run([os.path.join(ROOT, "bin", "helen_train"), "test", "--test_image_dir", str(d), "--model_path", model,
                        "--batch_size", "4", "--num_workers", "4", "--output_dir", str(tmp_path / "ev")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # (the loader walks the images in NAME order, so its batches hold other windows than the golden's: the expectation is
    # the pinned per-batch function over the loader's own batches; the confusion matrices do not depend on the order)
    from helen_amd.evaluate import SequenceDataset as Labeled
    order = [int(name.split("-")[1]) // 800 for _, name in Labeled(str(d)).all_images]
    assert sorted(order) == list(range(10)) and order != list(range(10))
    want_loss, scratch_b, scratch_r = 0.0, np.zeros((5, 5), np.int64), np.zeros((11, 11), np.int64)
    for lo in range(0, 10, 4):
        k = order[lo:lo + 4]
        a, b = host_batch_losses(e, img[k], g["label_base"][k], g["label_rle"][k], TrainOptions.CLASS_WEIGHTS, scratch_b, scratch_r)
        want_loss += float((a + b).sum())
    want_loss /= 10 * 19
    got = float(re.search(r"Test Loss: ([0-9.eE+-]+)", r.stderr).group(1))
    assert abs(got - want_loss) <= 1e-9 * want_loss, (got, want_loss)
    assert np.array_equal(np.loadtxt(str(tmp_path / "ev" / "BASE_CONFUSION_MATRIX.tsv"), dtype=np.int64), g["base_confusion_matrix"])
    assert np.array_equal(np.loadtxt(str(tmp_path / "ev" / "RLE_CONFUSION_MATRIX.tsv"), dtype=np.int64), g["rle_confusion_matrix"])
