"""TEST INFRASTRUCTURE -- generates tests/golden/predict_ref_large.npz by RUNNING the reference's own `predict` on
4,096 windows (8.19 M labels).

Runs only in the build container (needs /root/reference; tests/golden/reference_env.py documents what the image lacks
and how it is supplied):

    python tests/golden/make_golden_predict_large.py

Same executed function as make_golden_predict.py -- the reference's `helen/modules/python/models/predict.py:38-175`
(its `SequenceDataset` over an image DIRECTORY, torch's `DataLoader`, its `ModelHandler.load_simple_model`, its 19-chunk
loop with `TransducerGRU.forward`, softmax, zero-pad-add, argmax, its `DataStore.write_prediction`), on CPU, nothing
restated -- but on a sample large enough to put a rate on "HIP labels == reference labels": `large_case()` below, eight
image files of 512 seeded windows each (2,048 uniform, 2,048 pileup-like; one region per window, chunk id 0).  The images
are NOT stored: `large_case()` regenerates them from the seeds wherever the test runs.  The fixture holds what the
reference wrote: the two label datasets of every window, in window order, as uint8 [4096, 1000] arrays (compressed:
the labels of this peaked synthetic model are far from uniform), plus a digest of the rest of the tree (names, bounds,
positions).
"""
import hashlib
import io
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reference_env import ROOT, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "predict_ref_large.npz")
BATCH = 64
N_WINDOWS = 4096
N_FILES = 8
CONTIG = "chr20"


def large_images():
    sys.path.insert(0, ROOT)
    from helen_amd.weights import make_images
    return np.concatenate([make_images(N_WINDOWS // 2, seed=4101, mode="uniform"),
                           make_images(N_WINDOWS // 2, seed=4102, mode="pileup")])


def large_weights():
    sys.path.insert(0, ROOT)
    from helen_amd.weights import make_weights
    return make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0)


def region_of(i):
    """(file index, start, end) of window i: files hold contiguous runs, window i is region i of the contig."""
    return i // (N_WINDOWS // N_FILES), 800 * i, 800 * i + 1000


def large_case(directory, images=None):
    """The image directory and the model file.  -> (image_dir, model_path, images)"""
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    from helen_amd.model_handler import ModelHandler
    if images is None:
        images = large_images()
    image_dir = os.path.join(directory, "images")
    os.makedirs(image_dir)
    per = N_WINDOWS // N_FILES
    pos = np.zeros((1000, 3), np.int64)
    for k in range(N_FILES):
        with hdf5.File(os.path.join(image_dir, "pileups_%d.h5" % k), "w") as f:
            for i in range(k * per, (k + 1) * per):
                _, start, end = region_of(i)
                base = "images/%s-%d-%d-0/" % (CONTIG, start, end)
                f.write(base + "contig", CONTIG)
                f.write(base + "contig_start", np.array([start], np.int64))
                f.write(base + "contig_end", np.array([end], np.int64))
                f.write(base + "feature_chunk_idx", np.array([0], np.int64))
                f.write(base + "image", images[i], np.uint8)
                pos[:, 0] = start + np.arange(1000)
                f.write(base + "position", pos, np.int64)
    model = os.path.join(directory, "model.pkl")
    ModelHandler.save_model(large_weights(), None, 128, 1, 0, model)
    return image_dir, model, images


def labels_of(path):
    """(bases u8 [N,1000], rles u8 [N,1000], sha1 of everything else) of a prediction file written for large_case."""
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    bases = np.zeros((N_WINDOWS, 1000), np.uint8)
    rles = np.zeros((N_WINDOWS, 1000), np.uint8)
    seen = np.zeros(N_WINDOWS, bool)
    h = hashlib.sha1()
    with hdf5.File(path, "r") as f:
        assert f.keys("predictions") == [CONTIG], f.keys("predictions")
        regions = f.keys("predictions/" + CONTIG)
        for name in sorted(regions, key=lambda s: int(s.split("-")[-2])):
            start, end = int(name.split("-")[-2]), int(name.split("-")[-1])
            i = start // 800
            g = "predictions/%s/%s/" % (CONTIG, name)
            members = sorted(f.keys(g))
            assert members == ["0", "contig_end", "contig_start"], (name, members)
            assert sorted(f.keys(g + "0")) == ["bases", "position", "rles"], name
            bases[i] = f.read(g + "0/bases")
            rles[i] = f.read(g + "0/rles")
            seen[i] = True
            posn = f.read(g + "0/position")
            h.update(name.encode())
            for a in (f.read(g + "contig_start"), f.read(g + "contig_end"), posn):
                a = np.asarray(a)
                h.update(str((a.dtype.str, a.shape)).encode())
                h.update(np.ascontiguousarray(a).tobytes())
    assert seen.all(), "windows missing from the prediction file: %s" % np.flatnonzero(~seen)[:5]
    return bases, rles, h.hexdigest()


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference")
    install()
    sys.path.insert(0, "/root/reference")
    from helen.modules.python.models.predict import predict                   # the reference's own function
    d = tempfile.mkdtemp(prefix="helen_golden_predict_large_")
    stderr, sys.stderr = sys.stderr, io.StringIO()
    t0 = time.time()
    try:
        image_dir, model, _ = large_case(d)
        out = os.path.join(d, "reference_prediction.hdf")
        predict(image_dir, out, model, BATCH, 0, 8, False)
        import gc
        gc.collect()                                                            # the reference never closes its DataStore
        bases, rles, rest = labels_of(out)
    finally:
        log = sys.stderr.getvalue()
        sys.stderr = stderr
        shutil.rmtree(d, ignore_errors=True)
    np.savez_compressed(OUT, bases=bases, rles=rles, rest_sha1=np.array(rest), batch=np.array(BATCH),
                        made_by=np.array("tests/golden/make_golden_predict_large.py (reference models/predict.py "
                                         "executed on CPU, %d windows)" % N_WINDOWS))
    print(log[-300:])
    print("wrote %s: %d windows, %d bytes, %.0f s" % (OUT, N_WINDOWS, os.path.getsize(OUT), time.time() - t0))


if __name__ == "__main__":
    main()
