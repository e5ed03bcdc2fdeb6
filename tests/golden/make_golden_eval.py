#!/usr/bin/env python3
"""Golden vector for the evaluation path (`helen_train test`), generated with the REFERENCE model and
torch's own nn.CrossEntropyLoss -- build container only, like make_golden.py.

The loop below restates helen/modules/python/models/test.py:78-126 (that module itself needs torchnet,
which is not installed): zero hidden per batch, 19 chunks, CrossEntropyLoss() on the base logits +
CrossEntropyLoss(weight=CLASS_WEIGHTS) on the run-length logits per chunk, loss sums, total_images
advanced by the batch size per chunk, confusion[target][argmax] as torchnet's ConfusionMeter counts.

    python tests/golden/make_golden_eval.py       # writes tests/golden/eval10.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, load_reference_model, reference_batch  # noqa: E402,F401  (also sets sys.path)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from helen.modules.python.Options import ImageSizeOptions, TrainOptions  # noqa: E402  (reference)

from helen_amd.weights import make_images, make_weights  # noqa: E402

CASE = dict(seed=20260928, head_scale=8.0, input_scale=1.0 / 64.0)
N, BATCH = 10, 4


def case_inputs():
    """10 windows (7 uniform, 3 pileup-like); labels: the model's own predictions with a seeded third
    of the positions replaced by random classes (so both matrices have off-diagonal mass)."""
    img = np.concatenate([make_images(7, seed=31, mode="uniform"), make_images(3, seed=32, mode="pileup")])
    return img


def main():
    torch.set_num_threads(8)
    w = make_weights(**CASE)
    model = load_reference_model(w)
    img = case_inputs()
    pred = reference_batch(model, img)
    rng = np.random.default_rng(5)
    flip = rng.random((N, 1000)) < 0.33
    label_base = np.where(flip, rng.integers(0, 5, (N, 1000)), pred["bases"]).astype(np.uint8)
    label_rle = np.where(flip, rng.integers(0, 11, (N, 1000)), pred["rles"]).astype(np.uint8)

    criterion_base = nn.CrossEntropyLoss()
    criterion_rle = nn.CrossEntropyLoss(weight=torch.Tensor(TrainOptions.CLASS_WEIGHTS))
    conf_b = np.zeros((5, 5), np.int64)
    conf_r = np.zeros((11, 11), np.int64)
    total_loss = total_loss_rle = 0.0
    total_images = 0
    chunk_losses = []
    with torch.no_grad():
        for lo in range(0, N, BATCH):
            images = torch.from_numpy(img[lo:lo + BATCH]).type(torch.FloatTensor)
            lb = torch.from_numpy(label_base[lo:lo + BATCH]).type(torch.LongTensor)
            lr = torch.from_numpy(label_rle[lo:lo + BATCH]).type(torch.LongTensor)
            hidden = torch.zeros(images.size(0), 2 * TrainOptions.GRU_LAYERS, TrainOptions.HIDDEN_SIZE)
            for i in range(0, ImageSizeOptions.SEQ_LENGTH, TrainOptions.WINDOW_JUMP):
                if i + TrainOptions.TRAIN_WINDOW > ImageSizeOptions.SEQ_LENGTH:
                    break
                ob, orl, hidden = model(images[:, i:i + TrainOptions.TRAIN_WINDOW], hidden)
                lbc = lb[:, i:i + TrainOptions.TRAIN_WINDOW].contiguous().view(-1)
                lrc = lr[:, i:i + TrainOptions.TRAIN_WINDOW].contiguous().view(-1)
                loss_base = criterion_base(ob.contiguous().view(-1, 5), lbc)
                loss_rle = criterion_rle(orl.contiguous().view(-1, 11), lrc)
                total_loss += (loss_base + loss_rle).item()
                total_loss_rle += loss_rle.item()
                total_images += images.size(0)
                chunk_losses.append((loss_base.item(), loss_rle.item()))
                np.add.at(conf_b, (lbc.numpy(), ob.contiguous().view(-1, 5).numpy().argmax(1)), 1)
                np.add.at(conf_r, (lrc.numpy(), orl.contiguous().view(-1, 11).numpy().argmax(1)), 1)
    np.savez_compressed(os.path.join(HERE, "eval10.npz"), label_base=label_base, label_rle=label_rle,
                        loss=np.array([total_loss / total_images]), total_loss=np.array([total_loss]),
                        total_loss_rle=np.array([total_loss_rle]), total_images=np.array([total_images]),
                        chunk_losses=np.array(chunk_losses), base_confusion_matrix=conf_b,
                        rle_confusion_matrix=conf_r,
                        image_crc=np.array([int(img.astype(np.uint64).sum())], dtype=np.uint64))
    print("eval10 written: loss %.6f  total_loss_rle %.4f  base acc %.2f%%  rle acc %.2f%%"
          % (total_loss / total_images, total_loss_rle, 100.0 * np.trace(conf_b) / conf_b.sum(),
             100.0 * np.trace(conf_r) / conf_r.sum()))


if __name__ == "__main__":
    main()
