"""TEST INFRASTRUCTURE -- generates tests/golden/polish_ref.json.gz by RUNNING the reference's own polish chain.

Runs only in the build container (needs /root/reference and oracle/_ref/libssw_ref.so; tests/golden/reference_env.py
documents what the image lacks and how it is supplied):

    python tests/golden/make_golden_polish.py

What is executed is what `helen polish` chains (helen/modules/python/PolishInterface.py:49-105): the reference's
inference function `models/predict.py:35-175` (its SequenceDataset over the image DIRECTORY, torch DataLoader, its
ModelHandler on a `.pkl`, the 19-chunk loop with its TransducerGRU, softmax / zero-pad-add / argmax, its DataStore
writer) into `<dir>/predictions/<prefix>.hdf`, then its `StitchInterface.perform_stitch` (:40-106 -> Stitch.py:96-301:
per-region position dictionaries over the chunk ids in string order, runs of regions in worker processes, striped
Smith-Waterman joins on its own ssw.c) on that directory -- on CPU, nothing restated.  (`call_consensus` itself cannot
be imported here -- it pulls in onnxruntime -- and adds only argument checks and file sharding to this chain.)

Input: the simulated assembly helen_amd.synthetic.POLISH_CASE (four contigs in three image files: multi-image regions,
insert and split rows, short last images, a thirteen-image region, a hole, a contig shorter than one image) rendered
through the read-vote noise model, and the weights of tests/golden/trained_synth.npz (the reference model trained on
that noise model) saved in the reference's checkpoint format.  The fixture holds the reference's FASTA for 1 and 3
stitch threads, the prediction file as {dataset path: dtype, shape, SHA-1} and its label datasets (so that a test can
say WHICH labels differ should any).
"""
import base64
import gzip
import io
import json
import os
import shutil
import sys
import tempfile
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reference_env import ROOT, install  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "polish_ref.json.gz")
BATCH = 8
STITCH_THREADS = (1, 3)


def polish_case(directory, direct=False):
    """The image directory and the model file.  -> (image_dir, model_path, what write_assembly_dir returned)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_cases import load_trained_synth
    from helen_amd import synthetic as S
    from helen_amd.model_handler import ModelHandler
    image_dir = os.path.join(directory, "images")
    made = S.write_assembly_dir(image_dir, S.POLISH_CASE, S.POLISH_CASE_FILES, blocks=S.POLISH_CASE_BLOCKS, direct=direct)
    model = os.path.join(directory, "model.pkl")
    ModelHandler.save_model(load_trained_synth()[0], None, 128, 1, 0, model)
    return image_dir, model, made


def tree_and_labels(path):
    """({dataset path: dtype, shape, sha1}, [paths of the label datasets in tree order], their bytes joined)"""
    sys.path.insert(0, ROOT)
    from helen_amd import hdf5
    from make_golden_io import digest
    tree, label_paths, labels = {}, [], []
    with hdf5.File(path, "r") as f:
        def visit(group):
            for k in f.keys(group):
                p = group.rstrip("/") + "/" + k
                if k in ("position", "bases", "rles", "contig_start", "contig_end"):
                    a = f.read(p)
                    tree[p] = digest(a)
                    if k in ("bases", "rles"):
                        label_paths.append(p)
                        labels.append(np.ascontiguousarray(a).tobytes())
                else:
                    visit(p)
        visit("/")
    return tree, label_paths, b"".join(labels)


def read_fasta(path):
    with open(path) as f:
        return f.read()


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference")
    install()
    sys.path.insert(0, "/root/reference")
    from helen.modules.python.models.predict import predict                   # the reference's own functions
    from helen.modules.python.StitchInterface import perform_stitch
    d = tempfile.mkdtemp(prefix="helen_golden_polish_")
    stderr, sys.stderr = sys.stderr, io.StringIO()
    try:
        image_dir, model, made = polish_case(d)
        pred_dir = os.path.join(d, "predictions")
        os.makedirs(pred_dir)
        out = os.path.join(pred_dir, "polish_ref.hdf")
        predict(image_dir, out, model, BATCH, 0, 8, False)
        import gc
        gc.collect()                                                            # the reference never closes its DataStore
        tree, label_paths, labels = tree_and_labels(out)
        fasta = {}
        for t in STITCH_THREADS:
            perform_stitch(pred_dir, os.path.join(d, "fa%d" % t), "polished", t)
            fasta[str(t)] = read_fasta(os.path.join(d, "fa%d" % t, "polished.fa"))
    finally:
        log = sys.stderr.getvalue()
        sys.stderr = stderr
        shutil.rmtree(d, ignore_errors=True)
    with io.TextIOWrapper(gzip.GzipFile(OUT, "wb", mtime=0)) as f:
        json.dump({"made_by": "tests/golden/make_golden_polish.py (reference models/predict.py + StitchInterface.perform_stitch "
                              "executed on CPU)",
                   "batch": BATCH, "windows": made["windows"], "regions": made["regions"], "tree": tree,
                   "label_paths": label_paths, "labels_zb64": base64.b64encode(zlib.compress(labels, 9)).decode(),
                   "fasta": fasta, "stitch_warnings": [ln for ln in log.splitlines() if "WARNING" in ln]}, f)
    print(log[-600:])
    print("wrote %s: %d datasets, %d windows, FASTA %s bytes, fixture %d bytes; FASTA equal over thread counts: %s"
          % (OUT, len(tree), made["windows"], [len(v) for v in fasta.values()], os.path.getsize(OUT),
             len(set(fasta.values())) == 1))


if __name__ == "__main__":
    main()
