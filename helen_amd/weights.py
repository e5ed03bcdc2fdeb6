"""Deterministic, torch-free synthetic parameter set for the TransducerGRU.

No trained `.pkl` ships with the reference (they are downloaded, `DownloadModel.py:8-27`), so the
tests, the golden vectors and the benchmark all use this generator.  Names and shapes are the
reference model's `state_dict()` (`models/TransducerModel.py:43-58`); values follow PyTorch's
default init U(-1/sqrt(H), 1/sqrt(H)).  The two head weight matrices can be scaled so that the
softmax outputs are peaked like a trained model's ("Bit-identical argmax").
"""
import numpy as np

from .options import ImageSizeOptions, TrainOptions

def param_shapes(features=ImageSizeOptions.IMAGE_HEIGHT, hidden=TrainOptions.HIDDEN_SIZE):
    """(name, shape) list for a model with `features` inputs and `hidden` units."""
    g = 3 * hidden
    out = []
    for layer, k in (("gru_encoder", features), ("gru_decoder", 2 * hidden)):
        for suffix in ("", "_reverse"):
            out.append(("%s.weight_ih_l0%s" % (layer, suffix), (g, k)))
            out.append(("%s.weight_hh_l0%s" % (layer, suffix), (g, hidden)))
            out.append(("%s.bias_ih_l0%s" % (layer, suffix), (g,)))
            out.append(("%s.bias_hh_l0%s" % (layer, suffix), (g,)))
    out.append(("dense1_base.weight", (ImageSizeOptions.TOTAL_BASE_LABELS, 2 * hidden)))
    out.append(("dense1_base.bias", (ImageSizeOptions.TOTAL_BASE_LABELS,)))
    out.append(("dense2_rle.weight", (ImageSizeOptions.TOTAL_RLE_LABELS, 2 * hidden)))
    out.append(("dense2_rle.bias", (ImageSizeOptions.TOTAL_RLE_LABELS,)))
    return out


def make_weights(seed=20260928, head_scale=8.0, input_scale=1.0,
                 features=ImageSizeOptions.IMAGE_HEIGHT, hidden=TrainOptions.HIDDEN_SIZE):
    """Return an ordered dict name -> float32 ndarray.

    `input_scale` multiplies the encoder's input-to-hidden matrices.  The network is fed raw
    0..255 pileup counts (`models/predict_gpu.py:97`, no normalisation), so with U(-k, k) weights
    the encoder gates saturate; a trained model has learnt small input weights.  input_scale < 1
    (e.g. 1/64) reproduces that regime, which is the numerically harder one (gates un-saturated,
    state sensitive to rounding), so the parity tests use both.
    """
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(hidden)
    out = {}
    for name, shape in param_shapes(features, hidden):
        w = rng.uniform(-k, k, size=shape).astype(np.float32)
        if name in ("dense1_base.weight", "dense2_rle.weight"):
            w = (w * np.float32(head_scale)).astype(np.float32)
        if name.startswith("gru_encoder.weight_ih"):
            w = (w * np.float32(input_scale)).astype(np.float32)
        out[name] = w
    return out


PEAKED_HEAD_SCALE = 64.0


def make_peaked_weights(seed=20260928, input_scale=1.0 / 64.0):
    """The same random network with its two head matrices scaled by 64 instead of 8: the per-chunk softmax is then as
    sharp as a trained classifier's for most positions (median top-1 probability 0.999 for the base head and 0.90 for
    the run-length head on synthetic windows, against 0.60 / 0.30 at x8; DESIGN.md 5 gives the margin histogram).  No
    trained `.pkl` exists offline; this is the stand-in for "a trained, peaked model" when a reduced-precision mode's
    label identity or the fp32 tie rate is quoted.  The recurrent weights keep PyTorch's default-init range: scaling
    them saturates every gate and freezes the state, which is not what training does."""
    return make_weights(seed=seed, head_scale=PEAKED_HEAD_SCALE, input_scale=input_scale)


def margin_histogram(acc, edges=(2e-6, 1e-4, 1e-3, 1e-2, 1e-1)):
    """Fractions of positions whose top-1 / top-2 margin of the accumulated softmax (`acc` [..., classes], numpy) is
    below each edge -- 2e-6 is where two fp32 evaluations may legitimately call different labels."""
    a = np.sort(np.asarray(acc, dtype=np.float64), axis=-1)
    m = (a[..., -1] - a[..., -2]).ravel()
    return {"below_%g" % e: float((m < e).mean()) for e in edges}


def make_images(n_windows, seed=20260928, mode="uniform",
                features=ImageSizeOptions.IMAGE_HEIGHT, seq_length=ImageSizeOptions.SEQ_LENGTH):
    """Synthetic pileup windows, uint8 [n, seq_length, features].

    mode "uniform": i.i.d. uniform 0..255 (the benchmark's canonical input).
    mode "pileup":  about 4 of the 90 features non-zero per position, like a real MarginPolish
                    image (one dominant base/run-length/strand bucket plus noise).
    """
    rng = np.random.default_rng(seed)
    if mode == "uniform":
        return rng.integers(0, 256, size=(n_windows, seq_length, features), dtype=np.uint8)
    if mode == "pileup":
        img = np.zeros((n_windows, seq_length, features), dtype=np.uint8)
        n_hot = 4
        cols = rng.integers(0, features, size=(n_windows, seq_length, n_hot))
        vals = rng.integers(1, 256, size=(n_windows, seq_length, n_hot), dtype=np.uint8)
        w_idx = np.arange(n_windows)[:, None, None]
        p_idx = np.arange(seq_length)[None, :, None]
        img[w_idx, p_idx, cols] = vals
        return img
    raise ValueError("unknown image mode: " + str(mode))
