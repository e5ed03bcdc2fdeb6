"""TEST INFRASTRUCTURE -- trains the REFERENCE's own TransducerGRU on a synthetic polishing task and stores the weights.

Runs only in the build container (needs /root/reference and a few minutes of CPU):

    python tests/golden/make_trained_synth.py            # writes tests/golden/trained_synth.npz (4 epochs, ~100 s on 8 cores;
                                                         # regenerates byte-identical here: seeded, 8 torch threads)

No trained HELEN model exists offline (they are downloaded, helen/modules/python/DownloadModel.py:8-27), so every
other fixture and the benchmark use random-init weights, whose softmax outputs are flat and whose gates sit in their
linear range.  This script imports `helen.modules.python.models.TransducerModel.TransducerGRU` (nothing restated) and
trains it with torch autograd on helen_amd.synthetic.make_pileup_task -- windows with a known base and run-length label
per position and noisy read votes -- the way the reference trains (models/train.py:160-212: chunks of TRAIN_WINDOW
positions, hidden state carried across the chunks of a window and detached, CrossEntropyLoss on the base logits plus the
class-weighted one on the run-length logits, Adam).  The result is a network with TRAINED-looking parameters: large
structured input weights, saturating gates, confident outputs.  The fixture is data: the state dict (fp32), the
accuracy it reached, and the task's seed.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from helen.modules.python.models.TransducerModel import TransducerGRU  # noqa: E402  (reference)
from helen.modules.python.Options import ImageSizeOptions, TrainOptions  # noqa: E402  (reference)

from helen_amd.synthetic import make_pileup_task  # noqa: E402

OUT = os.path.join(HERE, "trained_synth.npz")
TASK_SEED = 7001
GOLDEN_WINDOWS = 8


def main():
    torch.manual_seed(20260929)
    torch.set_num_threads(8)
    model = TransducerGRU(ImageSizeOptions.IMAGE_CHANNELS, ImageSizeOptions.IMAGE_HEIGHT, TrainOptions.GRU_LAYERS,
                          TrainOptions.HIDDEN_SIZE, ImageSizeOptions.TOTAL_BASE_LABELS, ImageSizeOptions.TOTAL_RLE_LABELS)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    crit_base = nn.CrossEntropyLoss()
    crit_rle = nn.CrossEntropyLoss(weight=torch.tensor(TrainOptions.CLASS_WEIGHTS, dtype=torch.float32))
    W, J = TrainOptions.TRAIN_WINDOW, TrainOptions.WINDOW_JUMP
    t0 = time.time()
    steps = 0
    for epoch in range(int(os.environ.get("EPOCHS", "4"))):
        img, lb, lr = make_pileup_task(96, seed=TASK_SEED + epoch)
        x_all = torch.from_numpy(img).float()
        lb_t, lr_t = torch.from_numpy(lb).long(), torch.from_numpy(lr).long()
        for lo in range(0, x_all.shape[0], 32):
            x, yb, yr = x_all[lo:lo + 32], lb_t[lo:lo + 32], lr_t[lo:lo + 32]
            hidden = torch.zeros(x.shape[0], 2 * TrainOptions.GRU_LAYERS, TrainOptions.HIDDEN_SIZE)
            for i in range(0, ImageSizeOptions.SEQ_LENGTH - W + 1, J):
                ob, orl, hidden = model(x[:, i:i + W], hidden)
                loss = crit_base(ob.reshape(-1, 5), yb[:, i:i + W].reshape(-1)) + \
                    crit_rle(orl.reshape(-1, 11), yr[:, i:i + W].reshape(-1))
                opt.zero_grad()
                loss.backward()
                opt.step()
                hidden = hidden.detach()
                steps += 1
        print("epoch %d: %d steps, last loss %.4f, %.0f s" % (epoch, steps, float(loss), time.time() - t0), flush=True)
    # held-out accuracy through the reference's own sliding-window inference (predict_gpu.py:97-159 restated in torch)
    model.eval()
    img, lb, lr = make_pileup_task(32, seed=TASK_SEED + 1000)
    with torch.no_grad():
        x = torch.from_numpy(img).float()
        hidden = torch.zeros(x.shape[0], 2, TrainOptions.HIDDEN_SIZE)
        pb = torch.zeros(x.shape[0], 1000, 5)
        pr = torch.zeros(x.shape[0], 1000, 11)
        for i in range(0, 1000 - W + 1, J):
            ob, orl, hidden = model(x[:, i:i + W], hidden)
            pb[:, i:i + W] += torch.softmax(ob, 2)
            pr[:, i:i + W] += torch.softmax(orl, 2)
    acc_b = float((pb.argmax(2).numpy() == lb).mean())
    acc_r = float((pr.argmax(2).numpy() == lr).mean())
    print("held-out accuracy: base %.4f, run length %.4f" % (acc_b, acc_r))
    state = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    # what the reference's loop gives on GOLDEN_WINDOWS held-out windows (make_pileup_task(GOLDEN_WINDOWS, GOLDEN_SEED)):
    # labels and the accumulated softmax, for the oracle and the HIP path to be checked against on trained weights
    g = slice(0, GOLDEN_WINDOWS)
    np.savez_compressed(OUT, **state, _accuracy=np.array([acc_b, acc_r]), _task_seed=np.array(TASK_SEED),
                        _golden_seed=np.array(TASK_SEED + 1000), _ref_bases=pb[g].argmax(2).numpy().astype(np.uint8),
                        _ref_rles=pr[g].argmax(2).numpy().astype(np.uint8), _ref_acc_base=pb[g].numpy(),
                        _ref_acc_rle=pr[g].numpy(),
                        _made_by=np.array("tests/golden/make_trained_synth.py: reference TransducerGRU trained with "
                                          "torch autograd on helen_amd.synthetic.make_pileup_task, %d steps" % steps))
    print("wrote %s (%d bytes)" % (OUT, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
