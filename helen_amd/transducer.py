"""TransducerGRU: drop-in for the reference's model object at the operator boundary
(helen/modules/python/models/TransducerModel.py:20-93): same constructor, `forward(x, hidden)`,
`init_hidden`, `eval`, `to`, `state_dict` / `load_state_dict` -- but the forward is
libhelen_hip.so's helen_gru_chunk_forward on an MI355X.  The host executes it only on request: `use_cpu(threads)`
(what a run WITHOUT --gpu_mode does, like the reference's ONNX Runtime session, models/predict_cpu.py:39-170) binds the
model to libhelen_cpu.so; nothing ever falls back to it -- a model that was not told so needs a GPU.
"""
from collections import OrderedDict

import numpy as np
import torch

from .weights import param_shapes


class TransducerGRU(object):
    def __init__(self, image_channels, image_features, gru_layers, hidden_size, num_base_classes,
                 num_rle_classes, bidirectional=True):
        if gru_layers != 1 or not bidirectional:
            raise ValueError("this build implements the shipped HELEN architecture: one "
                             "bidirectional GRU layer per stage (Options.py:27)")
        self.hidden_size = hidden_size
        self.bidirectional = bidirectional
        self.num_layers = gru_layers
        self.num_base_classes = num_base_classes
        self.num_rle_classes = num_rle_classes
        self.image_features = image_features
        # parameters live on the host as float32 arrays until the model is bound to a device
        k = 1.0 / np.sqrt(hidden_size)
        rng = np.random.default_rng()
        self._params = OrderedDict(
            (name, rng.uniform(-k, k, size=shape).astype(np.float32))
            for name, shape in param_shapes(image_features, hidden_size))
        self._engine = None
        self._device = None
        self._max_windows = 4096
        # arithmetic of the gate matmuls: "fp32" (default, true fp32 MFMA), "fp32x3" (fp32-class via
        # three-term bf16 splits) or "bf16"; $HELEN_PRECISION selects it for the CLI
        import os
        self.precision = os.environ.get("HELEN_PRECISION", "fp32")
        self.training = False
        self._cpu_threads = None        # set by use_cpu(): the host engine, explicitly

    def use_cpu(self, threads=0):
        """Bind this model to the host path (libhelen_cpu.so) with `threads` OpenMP threads per call: the engine of a run
        without --gpu_mode (CallConsensusInterface.py:131,152).  Explicit: no other call ever selects it."""
        self._drop_engine()
        self._cpu_threads = int(threads)
        return self

    # ---- nn.Module-like surface used by the reference's callers ----
    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._params.items())

    def load_state_dict(self, state):
        missing = [k for k in self._params if k not in state]
        unexpected = [k for k in state if k not in self._params]
        if missing or unexpected:
            raise RuntimeError("Error(s) in loading state_dict for TransducerGRU: missing %s, "
                               "unexpected %s" % (missing, unexpected))
        for k in self._params:
            v = state[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != tuple(self._params[k].shape):
                raise RuntimeError("size mismatch for %s: %s vs %s"
                                   % (k, tuple(v.shape), tuple(self._params[k].shape)))
            self._params[k] = np.ascontiguousarray(v, dtype=np.float32)
        self._drop_engine()

    def eval(self):
        self.training = False
        return self

    def cpu(self):
        return self

    def to(self, device):
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if dev.type != "cuda":
            return self
        if self._device != dev:
            self._drop_engine()
            self._device = dev
        return self

    def cuda(self, device=0):
        return self.to(device)

    def set_capacity(self, max_windows):
        """Largest batch one call may carry (scratch is sized for it)."""
        if max_windows != self._max_windows:
            self._max_windows = int(max_windows)
            self._drop_engine()
        return self

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    @property
    def engine(self):
        """The device-resident replica (created on first use)."""
        if self._engine is None and self._cpu_threads is not None:
            from .cpu_engine import CpuEngine
            self._engine = CpuEngine(self._params, threads=self._cpu_threads)
        if self._engine is None:
            from .engine import HelenEngine
            dev = self._device if self._device is not None else torch.device("cuda", 0)
            self._engine = HelenEngine(self._params, device=dev.index or 0,
                                       max_windows=self._max_windows, precision=self.precision)
            self._device = dev
        return self._engine

    def forward(self, x, hidden):
        """x [B, T, 90] f32, hidden [B, 2, 128] -> (base [B,T,5], rle [B,T,11], hidden [B,2,128])
        (TransducerModel.py:60-79).  Inputs are moved to the model's device if needed."""
        if self._cpu_threads is not None:
            base, rle, h = self.engine.chunk_forward(x.detach().cpu().numpy(), hidden.detach().cpu().numpy())
            return torch.from_numpy(base), torch.from_numpy(rle), torch.from_numpy(h)
        dev = self._device if self._device is not None else torch.device("cuda", 0)
        if x.shape[0] > self._max_windows:
            self.set_capacity(x.shape[0])
        return self.engine.chunk_forward(x.to(dev), hidden.to(dev))

    __call__ = forward

    def init_hidden(self, batch_size, num_layers, bidirectional=True):
        num_directions = 2 if bidirectional else 1
        return torch.zeros(batch_size, num_directions * num_layers, self.hidden_size)
