"""ctypes wrapper around oracle/libhelen_oracle.so (plain-C restatement of the reference path).

TEST INFRASTRUCTURE ONLY -- see oracle/helen_oracle.c for what it follows and how it is pinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhelen_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


class HelenWeightsC(ctypes.Structure):
    """Mirror of `HelenWeights` in include/helen_hip.h."""
    _fields_ = [
        ("features", ctypes.c_int32), ("hidden", ctypes.c_int32),
        ("n_base", ctypes.c_int32), ("n_rle", ctypes.c_int32),
        ("enc_w_ih", _f32p * 2), ("enc_w_hh", _f32p * 2),
        ("enc_b_ih", _f32p * 2), ("enc_b_hh", _f32p * 2),
        ("dec_w_ih", _f32p * 2), ("dec_w_hh", _f32p * 2),
        ("dec_b_ih", _f32p * 2), ("dec_b_hh", _f32p * 2),
        ("base_w", _f32p), ("base_b", _f32p), ("rle_w", _f32p), ("rle_b", _f32p),
    ]


def build(force=False):
    """Compile the oracle with gcc if the .so is missing or older than its source."""
    src = os.path.join(_HERE, "helen_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "helen_hip.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhelen_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_gru_chunk_forward.restype = ctypes.c_int
        lib.oracle_gru_chunk_forward.argtypes = [
            ctypes.POINTER(HelenWeightsC), _f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p, _f32p,
            _f32p]
        lib.oracle_polish_batch.restype = ctypes.c_int
        lib.oracle_polish_batch.argtypes = [
            ctypes.POINTER(HelenWeightsC), _u8p, ctypes.c_int, _u8p, _u8p, _f32p, _f32p, _f32p,
            _f32p, _f32p]
        lib.oracle_polish_batch_f64.restype = ctypes.c_int
        lib.oracle_polish_batch_f64.argtypes = [ctypes.POINTER(HelenWeightsC), _u8p, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.oracle_max_threads.restype = ctypes.c_int
        lib.oracle_set_precision.argtypes = [ctypes.c_int]
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        _lib = lib
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


_NAMES = (
    ("enc_w_ih", "gru_encoder.weight_ih_l0"), ("enc_w_hh", "gru_encoder.weight_hh_l0"),
    ("enc_b_ih", "gru_encoder.bias_ih_l0"), ("enc_b_hh", "gru_encoder.bias_hh_l0"),
    ("dec_w_ih", "gru_decoder.weight_ih_l0"), ("dec_w_hh", "gru_decoder.weight_hh_l0"),
    ("dec_b_ih", "gru_decoder.bias_ih_l0"), ("dec_b_hh", "gru_decoder.bias_hh_l0"),
)


def weights_struct(weights):
    """dict name -> float32 ndarray (state_dict names) -> (HelenWeightsC, keepalive list)."""
    keep = []

    def arr(name):
        a = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
        keep.append(a)
        return a

    s = HelenWeightsC()
    w_ih = arr("gru_encoder.weight_ih_l0")
    w_hh = arr("gru_encoder.weight_hh_l0")
    s.features = w_ih.shape[1]
    s.hidden = w_hh.shape[1]
    s.n_base = arr("dense1_base.weight").shape[0]
    s.n_rle = arr("dense2_rle.weight").shape[0]
    for field, name in _NAMES:
        pair = getattr(s, field)
        pair[0] = _fp(arr(name))
        pair[1] = _fp(arr(name + "_reverse"))
    s.base_w = _fp(arr("dense1_base.weight"))
    s.base_b = _fp(arr("dense1_base.bias"))
    s.rle_w = _fp(arr("dense2_rle.weight"))
    s.rle_b = _fp(arr("dense2_rle.bias"))
    return s, keep


def max_threads():
    return int(_load().oracle_max_threads())


def set_threads(n):
    _load().oracle_set_threads(int(n))


def set_precision(name):
    """'fp32' (the reference arithmetic) or 'bf16' (gate-matmul operands rounded to bf16, fp32
    accumulate/state): the emulation the bf16 HIP variant is checked against."""
    _load().oracle_set_precision({"fp32": 0, "bf16": 1, "bf16_textbook": 2}[name])


def gru_chunk_forward(weights, x, h_in):
    """x [B,T,F] f32, h_in [B,2,H] f32 -> (base [B,T,5], rle [B,T,11], h_out [B,2,H])."""
    lib = _load()
    s, keep = weights_struct(weights)
    x = np.ascontiguousarray(x, dtype=np.float32)
    h_in = np.ascontiguousarray(h_in, dtype=np.float32)
    B, T, F = x.shape
    assert F == s.features and h_in.shape == (B, 2, s.hidden)
    base = np.empty((B, T, s.n_base), np.float32)
    rle = np.empty((B, T, s.n_rle), np.float32)
    h_out = np.empty((B, 2, s.hidden), np.float32)
    rc = lib.oracle_gru_chunk_forward(ctypes.byref(s), _fp(x), _fp(h_in), B, T, _fp(base),
                                      _fp(rle), _fp(h_out))
    if rc != 0:
        raise RuntimeError("oracle_gru_chunk_forward failed: %d" % rc)
    return base, rle, h_out


def polish_batch(weights, images, traces=False):
    """images [B,1000,F] u8 -> dict(bases, rles, acc_base, acc_rle[, hidden, logit_base, logit_rle])."""
    lib = _load()
    s, keep = weights_struct(weights)
    images = np.ascontiguousarray(images, dtype=np.uint8)
    B, L, F = images.shape
    assert L == 1000 and F == s.features
    out = {
        "bases": np.empty((B, L), np.uint8), "rles": np.empty((B, L), np.uint8),
        "acc_base": np.empty((B, L, s.n_base), np.float32),
        "acc_rle": np.empty((B, L, s.n_rle), np.float32),
    }
    null = ctypes.cast(None, _f32p)
    ht = lb = lr = null
    if traces:
        out["hidden"] = np.empty((19, B, 2, s.hidden), np.float32)
        out["logit_base"] = np.empty((19, B, 100, s.n_base), np.float32)
        out["logit_rle"] = np.empty((19, B, 100, s.n_rle), np.float32)
        ht, lb, lr = _fp(out["hidden"]), _fp(out["logit_base"]), _fp(out["logit_rle"])
    rc = lib.oracle_polish_batch(
        ctypes.byref(s), images.ctypes.data_as(_u8p), B, out["bases"].ctypes.data_as(_u8p),
        out["rles"].ctypes.data_as(_u8p), _fp(out["acc_base"]), _fp(out["acc_rle"]), ht, lb, lr)
    if rc != 0:
        raise RuntimeError("oracle_polish_batch failed: %d" % rc)
    return out


def polish_batch_f64(weights, images):
    """The float64 arbiter (oracle_polish_batch_f64): images [B,1000,F] u8 -> (acc_base f64 [B,1000,5],
    acc_rle f64 [B,1000,11]), the accumulated softmax of the same network evaluated in double precision end
    to end.  ~4 windows/s per thread: meant for the few windows on which two fp32 implementations disagree."""
    lib = _load()
    s, keep = weights_struct(weights)
    images = np.ascontiguousarray(images, dtype=np.uint8)
    B, L, F = images.shape
    assert L == 1000 and F == s.features
    ab = np.empty((B, L, s.n_base), np.float64)
    ar = np.empty((B, L, s.n_rle), np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.oracle_polish_batch_f64(ctypes.byref(s), images.ctypes.data_as(_u8p), B, ab.ctypes.data_as(dp),
                                     ar.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError("oracle_polish_batch_f64 failed: %d" % rc)
    return ab, ar


def arbitrate(weights, images, windows, kind, positions, labels_a, labels_b):
    """For label disagreements between two fp32 implementations A and B -- parallel arrays `windows` (index into
    `images`), `kind` ('bases' | 'rles'), `positions`, `labels_a`, `labels_b` -- evaluate those windows in float64
    and return one dict per disagreement: the float64 argmax, the float64 top-1 / top-2 margin and the float64
    gap between the two contested classes."""
    uniq = sorted(set(int(w) for w in windows))
    ab, ar = polish_batch_f64(weights, np.ascontiguousarray(images[uniq]))
    at = {w: i for i, w in enumerate(uniq)}
    out = []
    for w, k, p, la, lb in zip(windows, kind, positions, labels_a, labels_b):
        acc = (ab if k == "bases" else ar)[at[int(w)], int(p)]
        srt = np.sort(acc)
        out.append({"window": int(w), "kind": k, "position": int(p), "a": int(la), "b": int(lb),
                    "f64_argmax": int(acc.argmax()), "f64_margin": float(srt[-1] - srt[-2]),
                    "f64_gap_a_b": float(abs(acc[int(la)] - acc[int(lb)]))})
    return out


def evaluate(weights, images, label_base, label_rle, batch_size, class_weights):
    """The reference's evaluation loop (helen/modules/python/models/test.py:78-126, 150) on the CPU
    restatement's per-chunk logits: per loader batch and chunk nn.CrossEntropyLoss (mean) on the base
    logits + class-weighted nn.CrossEntropyLoss (sum w*nll / sum w) on the run-length logits, and
    torchnet ConfusionMeter counts conf[target][argmax(logits)].  float64 bookkeeping."""
    images = np.ascontiguousarray(images, np.uint8)
    n = images.shape[0]
    cw = np.asarray(class_weights, np.float64)
    total_loss = total_loss_rle = 0.0
    total_images = 0
    conf_b = np.zeros((5, 5), np.int64)
    conf_r = np.zeros((11, 11), np.int64)
    chunk_losses = []

    def nll(logits, labels):
        x = logits.astype(np.float64)
        m = x.max(axis=-1, keepdims=True)
        lse = m[..., 0] + np.log(np.exp(x - m).sum(axis=-1))
        return lse - np.take_along_axis(x, labels[..., None].astype(np.int64), axis=-1)[..., 0]

    for lo in range(0, n, batch_size):
        hi = min(n, lo + batch_size)
        tr = polish_batch(weights, images[lo:hi], traces=True)
        for c in range(19):
            lb = label_base[lo:hi, 50 * c:50 * c + 100]
            lr = label_rle[lo:hi, 50 * c:50 * c + 100]
            ob, orl = tr["logit_base"][c], tr["logit_rle"][c]          # [B,100,C]
            loss_b = nll(ob, lb).mean()
            w = cw[lr.astype(np.int64)]
            loss_r = (w * nll(orl, lr)).sum() / w.sum()
            total_loss += loss_b + loss_r
            total_loss_rle += loss_r
            total_images += hi - lo
            chunk_losses.append((loss_b, loss_r))
            np.add.at(conf_b, (lb.astype(np.int64).ravel(), ob.argmax(axis=-1).ravel()), 1)
            np.add.at(conf_r, (lr.astype(np.int64).ravel(), orl.argmax(axis=-1).ravel()), 1)
    return {"loss": total_loss / total_images if total_images else 0.0, "total_loss": total_loss,
            "total_loss_rle": total_loss_rle, "total_images": total_images,
            "base_confusion_matrix": conf_b, "rle_confusion_matrix": conf_r,
            "chunk_losses": np.array(chunk_losses)}
