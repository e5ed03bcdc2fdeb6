"""Evaluation of a model on labeled images: the `helen_train test` path.

Mirrors helen/modules/python/models/test.py:17-167 and TestInterface.py:93-141: same arguments, same
returned dictionary, same loss bookkeeping -- per loader batch and chunk a CrossEntropyLoss mean over
the base logits plus a class-weighted CrossEntropyLoss over the run-length logits, summed into
`total_loss`, divided at the end by `total_images` (which the reference advances by the batch size
once per CHUNK), and torchnet-style confusion matrices [target][predicted] of every chunk's argmax.
The arithmetic runs on the MI355X (helen_evaluate_batch): the 19-chunk forward is the polish path's,
the heads kernel emits the cross-entropy terms and confusion counts instead of softmax accumulators.
Loader batches are coalesced per device call; the per-batch means are finished on the host from the
kernel's per-window partial sums, so the loss does not depend on the coalescing.
"""
import sys

import numpy as np

from . import hdf5, native_io
from .file_manager import get_file_paths_from_directory
from .options import ImageSizeOptions, TrainOptions

DEVICE_CALL_WINDOWS = 4096


class SequenceDataset(object):
    """Labeled images (models/dataloader.py:11-70): items are (image, label_base, label_run_length)
    exactly as stored -- this loader does not pad, so every image must already be SEQ_LENGTH long
    (torch's default collate would refuse a ragged batch)."""

    def __init__(self, image_directory):
        pairs = []
        for path in get_file_paths_from_directory(image_directory):
            if native_io.available():
                names = native_io.list_images(path)
            else:
                with hdf5.File(path, "r") as f:
                    names = f.keys("images") if "images" in f else None
            if names is None:
                sys.stderr.write("WARN: NO IMAGES FOUND IN FILE: " + path + "\n")
            else:
                pairs.extend((path, name) for name in names)
        self.all_images = pairs
        self._files = {}

    def __len__(self):
        return len(self.all_images)

    def _file(self, path):
        f = self._files.get(path)
        if f is None:
            if len(self._files) >= 64:
                self._files.popitem()[1].close()
            f = self._files[path] = hdf5.File(path, "r")
        return f

    def __getitem__(self, index):
        path, name = self.all_images[index]
        f = self._file(path)
        base = "images/" + name + "/"
        return (f.read(base + "image", np.uint8), f.read(base + "label_base", np.uint8),
                f.read(base + "label_run_length", np.uint8))

    def read_range(self, lo, hi):
        """Items lo..hi-1 stacked: images u8 [n,1000,90], label_base / label_rle u8 [n,1000]."""
        n = hi - lo
        L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        images = np.empty((n, L, H), np.uint8)
        lb = np.empty((n, L), np.uint8)
        lr = np.empty((n, L), np.uint8)
        if native_io.available():
            # whole runs of one file per call, through the direct scanner (three dataset opens per image through the
            # ctypes binding are 0.3 ms: 3 k images/s against 80 k/s of device throughput)
            k = 0
            while k < n:
                path = self.all_images[lo + k][0]
                e = k
                while e < n and self.all_images[lo + e][0] == path:
                    e += 1
                native_io.read_labeled(path, [name for _, name in self.all_images[lo + k:lo + e]], images[k:e],
                                       lb[k:e], lr[k:e])
                k = e
        for k in range(0 if not native_io.available() else n, n):
            image, b, r = self[lo + k]
            if image.shape != (L, H) or b.shape != (L,) or r.shape != (L,):
                raise ValueError("IMAGE SIZE ERROR: " + str(self.all_images[lo + k][0]) + " "
                                 + str(image.shape) + " " + str(b.shape) + " " + str(r.shape))
            images[k], lb[k], lr[k] = image, b, r
        if lb.max(initial=0) >= ImageSizeOptions.TOTAL_BASE_LABELS or \
                lr.max(initial=0) >= ImageSizeOptions.TOTAL_RLE_LABELS:
            raise ValueError("LABEL OUT OF RANGE IN " + str(self.all_images[lo][0]))    # torch: IndexError
        return images, lb, lr


def batch_losses(stats, batch_sizes):
    """Per loader batch and chunk, from the kernel's partial sums stats [n,19,10,3] (float32):
    loss_base = sum(nll)/(B*100) (nn.CrossEntropyLoss mean), loss_rle = sum(w*nll)/sum(w)
    (weighted mean).  Returns (loss_base, loss_rle) arrays [n_batches, 19], float64."""
    s = stats.astype(np.float64).sum(axis=2)                    # [n,19,3]
    lb, lr, lo = [], [], 0
    for b in batch_sizes:
        part = s[lo:lo + b].sum(axis=0)                         # [19,3]
        lb.append(part[:, 0] / (b * TrainOptions.TRAIN_WINDOW))
        lr.append(part[:, 1] / part[:, 2])
        lo += b
    return np.array(lb), np.array(lr)


def test(data_filepath, batch_size, gpu_mode, transducer_model, num_workers, gru_layers, hidden_size,
         num_base_classes, num_rle_classes, print_details=False):
    """models/test.py:17-167.  `transducer_model` is helen_amd.transducer.TransducerGRU (already on its
    device); returns {'loss', 'accuracy', 'base_confusion_matrix', 'rle_confusion_matrix'}."""
    import torch
    if not gpu_mode:
        return _test_on_host(data_filepath, batch_size, transducer_model, num_workers, num_base_classes, num_rle_classes)
    test_data = SequenceDataset(data_filepath)
    transducer_model.eval()
    class_weights = np.array(TrainOptions.CLASS_WEIGHTS, np.float32)
    group = max(1, DEVICE_CALL_WINDOWS // batch_size) * batch_size
    transducer_model.set_capacity(min(DEVICE_CALL_WINDOWS, group))
    engine = transducer_model.engine
    dev = engine.device
    base_cm = torch.zeros((num_base_classes, num_base_classes), dtype=torch.int64, device=dev)
    rle_cm = torch.zeros((num_rle_classes, num_rle_classes), dtype=torch.int64, device=dev)
    sys.stderr.write("Test starting\n")
    total_loss = total_loss_rle = 0.0
    total_images = 0
    accuracy = 0
    n = len(test_data)
    with torch.cuda.device(dev):
        for lo in range(0, n, group):
            hi = min(n, lo + group)
            images, lb, lr = test_data.read_range(lo, hi)
            stats = engine.evaluate(torch.from_numpy(images).to(dev), torch.from_numpy(lb).to(dev),
                                    torch.from_numpy(lr).to(dev), class_weights, base_cm, rle_cm)
            sizes = [min(batch_size, hi - s) for s in range(lo, hi, batch_size)]
            loss_b, loss_r = batch_losses(stats.cpu().numpy(), sizes)
            for bi, b in enumerate(sizes):          # models/test.py:113-125, per chunk
                total_loss += float((loss_b[bi] + loss_r[bi]).sum())
                total_loss_rle += float(loss_r[bi].sum())
                total_images += b * loss_b.shape[1]
            bc, rc = base_cm.cpu().numpy(), rle_cm.cpu().numpy()
            sys.stderr.write("Base acc: %s, RLE acc: %s, RLE loss: %s\n" % (
                round(100.0 * np.trace(bc) / max(1.0, bc.sum()), 4),
                round(100.0 * np.trace(rc) / max(1.0, rc.sum()), 4), round(total_loss_rle, 4)))
    avg_loss = total_loss / total_images if total_images else 0
    bc, rc = base_cm.cpu().numpy(), rle_cm.cpu().numpy()
    sys.stderr.write("\nTest Loss: " + str(avg_loss) + "\n")
    sys.stderr.write("Base Confusion Matrix: \n" + str(bc) + "\n")
    sys.stderr.write("RLE Confusion Matrix: \n")
    for row in rc:
        sys.stderr.write("".join("{:9d} ".format(int(e)) for e in row) + "\n")
    return {"loss": avg_loss, "accuracy": accuracy, "base_confusion_matrix": bc,
            "rle_confusion_matrix": rc, "total_loss_rle": total_loss_rle, "total_images": total_images}


def host_batch_losses(engine, images, label_base, label_rle, class_weights, base_cm, rle_cm):
    """models/test.py:78-126 for ONE loader batch on the host engine (helen_amd.cpu_engine.CpuEngine): hidden = 0, then per
    chunk the logits of TransducerGRU.forward, nn.CrossEntropyLoss (mean) on the base logits and the class-weighted one
    (weighted mean, Options.py:29) on the run-length logits -- in float64 from fp32 logits -- and the confusion counts
    [target][argmax].  -> (loss_base [19], loss_rle [19])."""
    n = images.shape[0]
    x = images.astype(np.float32)
    h = np.zeros((n, 2, TrainOptions.HIDDEN_SIZE), np.float32)
    w = np.asarray(class_weights, np.float64)
    lb_out, lr_out = [], []
    rows = np.arange(n)[:, None]
    for c in range(19):
        lo = c * TrainOptions.WINDOW_JUMP
        base, rle, h = engine.chunk_forward(x[:, lo:lo + TrainOptions.TRAIN_WINDOW], h)
        tb = label_base[:, lo:lo + TrainOptions.TRAIN_WINDOW].astype(np.int64)
        tr = label_rle[:, lo:lo + TrainOptions.TRAIN_WINDOW].astype(np.int64)
        cols = np.arange(tb.shape[1])[None, :]

        def nll(logits, target):
            z = logits.astype(np.float64)
            z = z - z.max(-1, keepdims=True)
            return -(z[rows, cols, target] - np.log(np.exp(z).sum(-1)))
        lb_out.append(nll(base, tb).mean())
        wr = w[tr]
        lr_out.append((nll(rle, tr) * wr).sum() / wr.sum())
        np.add.at(base_cm, (tb.ravel(), base.argmax(-1).ravel()), 1)
        np.add.at(rle_cm, (tr.ravel(), rle.argmax(-1).ravel()), 1)
    return np.array(lb_out), np.array(lr_out)


def _test_on_host(data_filepath, batch_size, transducer_model, num_workers, num_base_classes, num_rle_classes):
    """`helen_train test` WITHOUT --gpu_mode (the reference evaluates on the CPU then, models/test.py:56-60): the same
    loop on the host engine, `num_workers` threads at most."""
    import os
    test_data = SequenceDataset(data_filepath)
    transducer_model.eval()
    threads = max(1, min(int(num_workers) if num_workers else 1, os.cpu_count() or 1))
    # a host engine of its own from the model's parameters: the caller's model object is left as it was (a later
    # test(..., gpu_mode=True) or forward() on it must not find itself on the host path)
    from .cpu_engine import CpuEngine
    engine = CpuEngine({k: v.detach().cpu().numpy() for k, v in transducer_model.state_dict().items()}, threads=threads)
    base_cm = np.zeros((num_base_classes, num_base_classes), np.int64)
    rle_cm = np.zeros((num_rle_classes, num_rle_classes), np.int64)
    sys.stderr.write("Test starting (host path, %d threads)\n" % threads)
    total_loss = total_loss_rle = 0.0
    total_images = 0
    n = len(test_data)
    for lo in range(0, n, batch_size):
        hi = min(n, lo + batch_size)
        images, lb, lr = test_data.read_range(lo, hi)
        loss_b, loss_r = host_batch_losses(engine, images, lb, lr, TrainOptions.CLASS_WEIGHTS, base_cm, rle_cm)
        total_loss += float((loss_b + loss_r).sum())
        total_loss_rle += float(loss_r.sum())
        total_images += (hi - lo) * len(loss_b)
        sys.stderr.write("Base acc: %s, RLE acc: %s, RLE loss: %s\n" % (
            round(100.0 * np.trace(base_cm) / max(1.0, base_cm.sum()), 4),
            round(100.0 * np.trace(rle_cm) / max(1.0, rle_cm.sum()), 4), round(total_loss_rle, 4)))
    avg_loss = total_loss / total_images if total_images else 0
    sys.stderr.write("\nTest Loss: " + str(avg_loss) + "\n")
    return {"loss": avg_loss, "accuracy": 0, "base_confusion_matrix": base_cm, "rle_confusion_matrix": rle_cm,
            "total_loss_rle": total_loss_rle, "total_images": total_images}


def test_interface(test_file, batch_size, gpu_mode, num_workers, model_path, output_directory,
                   print_details):
    """TestInterface.py:93-141: load the model, evaluate, save both confusion matrices under
    `output_directory` (as .tsv: the reference draws them with matplotlib)."""
    import os

    from .file_manager import handle_output_directory
    from .model_handler import ModelHandler
    sys.stderr.write("Loading data\n")
    output_directory = handle_output_directory(output_directory)
    if os.path.isfile(model_path) is False:
        sys.stderr.write("ERROR: INVALID PATH TO MODEL\n")
        sys.exit(1)
    sys.stderr.write("INFO: MODEL LOADING\n")
    transducer_model, hidden_size, gru_layers, prev_ite = ModelHandler.load_simple_model(
        model_path, input_channels=ImageSizeOptions.IMAGE_CHANNELS,
        image_features=ImageSizeOptions.IMAGE_HEIGHT, seq_len=ImageSizeOptions.SEQ_LENGTH,
        num_base_classes=ImageSizeOptions.TOTAL_BASE_LABELS,
        num_rle_classes=ImageSizeOptions.TOTAL_RLE_LABELS)
    sys.stderr.write("INFO: MODEL LOADED\n")
    if gpu_mode:
        transducer_model = transducer_model.cuda()
    stats = test(test_file, batch_size, gpu_mode, transducer_model, num_workers, gru_layers, hidden_size,
                 num_base_classes=ImageSizeOptions.TOTAL_BASE_LABELS,
                 num_rle_classes=ImageSizeOptions.TOTAL_RLE_LABELS, print_details=print_details)
    np.savetxt(os.path.join(output_directory, "RLE_CONFUSION_MATRIX.tsv"),
               stats["rle_confusion_matrix"], fmt="%d", delimiter="\t")
    np.savetxt(os.path.join(output_directory, "BASE_CONFUSION_MATRIX.tsv"),
               stats["base_confusion_matrix"], fmt="%d", delimiter="\t")
    sys.stderr.write("DONE\n")
    return stats
