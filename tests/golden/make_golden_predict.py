"""TEST INFRASTRUCTURE -- generates tests/golden/predict_ref.json.gz by RUNNING the reference's own `predict`.

Runs only in the build container (needs /root/reference; tests/golden/reference_env.py documents what the image lacks
and how it is supplied):

    python tests/golden/make_golden_predict.py

What is executed is the reference's whole inference function, `helen/modules/python/models/predict.py:38-175`
(`predict(test_file, output_filename, model_path, batch_size, num_workers, threads, gpu_mode=False)`): its
`SequenceDataset` over an image DIRECTORY, torch's `DataLoader` (batch 4, default collate, short last batch), its
`ModelHandler.load_simple_model` on a `.pkl`, its 19-chunk sliding-window loop with `TransducerGRU.forward`, softmax,
zero-pad-add and argmax, and its `DataStore.write_prediction` -- on CPU, nothing restated.  Inputs: `predict_case()`
below (two image files, 22 windows, short images, several regions and chunk ids; seeded) and this repository's
deterministic synthetic weights saved in the reference's checkpoint format.  The fixture stores the prediction file the
reference wrote, as a tree {dataset path: dtype, shape, SHA-1}, with the label datasets also kept whole (base64), so that
a test can say WHICH labels differ should any.
"""
import base64
import gzip
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reference_env import ROOT, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "predict_ref.json.gz")
BATCH = 4


def predict_case(directory):
    """The image directory and the model file.  -> (image_dir, model_path)"""
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    from helen_amd.model_handler import ModelHandler
    from helen_amd.weights import make_images, make_weights
    image_dir = os.path.join(directory, "images")
    os.makedirs(image_dir)
    img = np.concatenate([make_images(11, seed=41, mode="uniform"), make_images(11, seed=42, mode="pileup")])
    lengths = [1000] * 22
    lengths[3], lengths[9], lengths[17] = 613, 1, 999
    for k, lo, hi in ((0, 0, 12), (1, 12, 22)):
        with hdf5.File(os.path.join(image_dir, "pileups_%d.h5" % k), "w") as f:
            for i in range(lo, hi):
                L = lengths[i]
                region, chunk = i // 2, i % 2
                start = 800 * region
                contig = "chr20" if i < 16 else "chrUn_scaffold.7"
                base = "images/%s-%d-%d-%d/" % (contig, start, start + 1000, chunk)
                f.write(base + "contig", contig)
                f.write(base + "contig_start", np.array([start], np.int64))
                f.write(base + "contig_end", np.array([start + 1000], np.int64))
                f.write(base + "feature_chunk_idx", np.array([chunk], np.int64))
                f.write(base + "image", img[i, :L], np.uint8)
                pos = np.zeros((L, 3), np.int64)
                pos[:, 0] = start + np.arange(L) // 2
                pos[:, 1] = np.arange(L) % 2
                f.write(base + "position", pos, np.int64)
    model = os.path.join(directory, "model.pkl")
    ModelHandler.save_model(make_weights(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0), None, 128, 1, 0, model)
    return image_dir, model


def tree_of(path):
    """{dataset path: dtype, shape, sha1 (+ b64 for the label datasets)} of a prediction file."""
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    from make_golden_io import digest, walk  # noqa: F401
    out = {}
    with hdf5.File(path, "r") as f:
        def visit(group):
            for k in f.keys(group):
                p = group.rstrip("/") + "/" + k
                if k in ("position", "bases", "rles", "contig_start", "contig_end"):
                    a = f.read(p)
                    out[p] = digest(a)
                    if k in ("bases", "rles"):
                        out[p]["b64"] = base64.b64encode(np.ascontiguousarray(a).tobytes()).decode()
                else:
                    visit(p)
        visit("/")
    return out


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference")
    install()
    sys.path.insert(0, "/root/reference")
    from helen.modules.python.models.predict import predict                   # the reference's own function
    d = tempfile.mkdtemp(prefix="helen_golden_predict_")
    stderr, sys.stderr = sys.stderr, io.StringIO()
    try:
        image_dir, model = predict_case(d)
        out = os.path.join(d, "reference_prediction.hdf")
        predict(image_dir, out, model, BATCH, 0, 8, False)
        import gc
        gc.collect()                                                            # the reference never closes its DataStore
        tree = tree_of(out)
    finally:
        log = sys.stderr.getvalue()
        sys.stderr = stderr
        shutil.rmtree(d, ignore_errors=True)
    with io.TextIOWrapper(gzip.GzipFile(OUT, "wb", mtime=0)) as f:
        json.dump({"made_by": "tests/golden/make_golden_predict.py (reference models/predict.py executed on CPU)",
                   "batch": BATCH, "tree": tree}, f)
    print(log[-300:])
    print("wrote %s: %d datasets, %d bytes" % (OUT, len(tree), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
