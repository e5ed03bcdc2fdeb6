#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE model.

Runs only in the build container (it imports /root/reference, which never travels to the GPU
box).  It imports the reference's own `TransducerGRU` (helen/modules/python/models/
TransducerModel.py) and `ModelHandler.load_simple_model` (models/ModelHander.py:38-82), feeds
them this repo's deterministic synthetic weights saved in the reference's checkpoint format
(ModelHander.py:127-133), and drives them with the batch loop of models/predict_gpu.py:97-159
restated in torch on CPU.  The committed outputs are data only: inputs are regenerated from
seeds by helen_amd.weights, and the expected outputs are stored here.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from helen.modules.python.models.ModelHander import ModelHandler  # noqa: E402  (reference)
from helen.modules.python.Options import ImageSizeOptions, TrainOptions  # noqa: E402  (reference)

from helen_amd.weights import make_images, make_weights  # noqa: E402


def load_reference_model(weights):
    """Save `weights` as a reference-format .pkl and load it through the reference's loader."""
    state = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synthetic.pkl")
        torch.save({"model_state_dict": state, "model_optimizer": {}, "hidden_size": 128,
                    "gru_layers": 1, "epochs": 0}, path)
        model, hidden_size, gru_layers, _ = ModelHandler.load_simple_model(
            path, input_channels=ImageSizeOptions.IMAGE_CHANNELS,
            image_features=ImageSizeOptions.IMAGE_HEIGHT, seq_len=ImageSizeOptions.SEQ_LENGTH,
            num_base_classes=ImageSizeOptions.TOTAL_BASE_LABELS,
            num_rle_classes=ImageSizeOptions.TOTAL_RLE_LABELS)
    assert hidden_size == 128 and gru_layers == 1
    model.eval()
    return model


def reference_batch(model, images_u8, traces=False):
    """models/predict_gpu.py:97-159 on CPU; returns labels, accumulators and optional traces."""
    out = {"hidden": [], "logit_base": [], "logit_rle": []}
    with torch.no_grad():
        images = torch.from_numpy(images_u8).type(torch.FloatTensor)
        hidden = torch.zeros(images.size(0), 2 * TrainOptions.GRU_LAYERS, TrainOptions.HIDDEN_SIZE)
        pb = torch.zeros((images.size(0), images.size(1), ImageSizeOptions.TOTAL_BASE_LABELS))
        pr = torch.zeros((images.size(0), images.size(1), ImageSizeOptions.TOTAL_RLE_LABELS))
        for i in range(0, ImageSizeOptions.SEQ_LENGTH, TrainOptions.WINDOW_JUMP):
            if i + TrainOptions.TRAIN_WINDOW > ImageSizeOptions.SEQ_LENGTH:
                break
            chunk = images[:, i:i + TrainOptions.TRAIN_WINDOW]
            ob, orl, hidden = model(chunk, hidden)
            top, bottom = i, ImageSizeOptions.SEQ_LENGTH - (i + TrainOptions.TRAIN_WINDOW)
            layers = nn.Sequential(nn.Softmax(dim=2), nn.ZeroPad2d((0, 0, top, bottom)))
            pb = torch.add(pb, layers(ob))
            pr = torch.add(pr, layers(orl))
            if traces:
                out["hidden"].append(hidden.numpy().copy())
                out["logit_base"].append(ob.numpy().copy())
                out["logit_rle"].append(orl.numpy().copy())
        _, bl = torch.max(pb, 2)
        _, rl = torch.max(pr, 2)
    res = {"bases": bl.numpy().astype(np.uint8), "rles": rl.numpy().astype(np.uint8),
           "acc_base": pb.numpy(), "acc_rle": pr.numpy()}
    if traces:
        for k in ("hidden", "logit_base", "logit_rle"):
            res[k] = np.stack(out[k])
    return res


def case_images(case):
    """Seeded inputs of one golden case (tests regenerate these; they are not stored)."""
    if case == "trace6":
        # 4 uniform windows, 1 pileup-like sparse window, 1 short window padded with zero rows as
        # SequenceDataset does (dataloader_predict.py:74-82).
        img = np.concatenate([make_images(4, seed=11, mode="uniform"),
                              make_images(2, seed=12, mode="pileup")])
        img[5, 613:, :] = 0
        return img
    if case == "small_input6":
        return case_images("trace6")
    if case == "config1_100":
        # BASELINE.json configs[0]: 100 synthetic windows, batch 4 (F=90: Options.py:14).
        return np.concatenate([make_images(60, seed=21, mode="uniform"),
                               make_images(40, seed=22, mode="pileup")])
    raise KeyError(case)


CASE_WEIGHTS = {
    "trace6": dict(seed=20260928, head_scale=8.0, input_scale=1.0),
    "small_input6": dict(seed=7, head_scale=8.0, input_scale=1.0 / 64.0),
    "config1_100": dict(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0),
}


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    for case in ("trace6", "small_input6"):
        w = make_weights(**CASE_WEIGHTS[case])
        model = load_reference_model(w)
        img = case_images(case)
        r = reference_batch(model, img, traces=True)
        # also one direct TransducerGRU.forward call with a non-zero hidden and T < 100
        rng = np.random.default_rng(99)
        x = rng.integers(0, 256, size=(3, 37, 90)).astype(np.float32)
        h0 = rng.uniform(-1, 1, size=(3, 2, 128)).astype(np.float32)
        with torch.no_grad():
            fb, fr, fh = model(torch.from_numpy(x), torch.from_numpy(h0))
        np.savez_compressed(
            os.path.join(HERE, case + ".npz"),
            bases=r["bases"], rles=r["rles"],
            acc_base=r["acc_base"][:3], acc_rle=r["acc_rle"][:3],          # first 3 windows
            hidden=r["hidden"],                                             # [19,6,2,128]
            logit_base=r["logit_base"][[0, 9, 18]], logit_rle=r["logit_rle"][[0, 9, 18]],
            fwd_x=x, fwd_h0=h0, fwd_base=fb.numpy(), fwd_rle=fr.numpy(), fwd_h=fh.numpy(),
            image_crc=np.array([int(img.astype(np.uint64).sum())], dtype=np.uint64))
        print(case, "written; min top1-top2 margin base/rle:",
              margins(r["acc_base"]), margins(r["acc_rle"]))

    # BASELINE.json configs[0]: 100 windows, batch 4, reference PyTorch CPU path; timed.
    case = "config1_100"
    w = make_weights(**CASE_WEIGHTS[case])
    model = load_reference_model(w)
    img = case_images(case)
    bases, rles = [], []
    t0 = time.time()
    for s in range(0, 100, 4):
        r = reference_batch(model, img[s:s + 4])
        bases.append(r["bases"])
        rles.append(r["rles"])
    dt = time.time() - t0
    np.savez_compressed(os.path.join(HERE, case + ".npz"), bases=np.concatenate(bases),
                        rles=np.concatenate(rles),
                        image_crc=np.array([int(img.astype(np.uint64).sum())], dtype=np.uint64))
    print(case, "written; reference torch CPU, batch 4, 8 threads: %.1f s = %.1f windows/s"
          % (dt, 100 / dt))


def margins(acc):
    s = np.sort(acc, axis=2)
    return float((s[:, :, -1] - s[:, :, -2]).min())


if __name__ == "__main__":
    main()
