"""helen_amd: the `helen polish` RNN inference path, MI355X-native.

Only the hot path of kishwarshafin/helen is here: pileup windows -> TransducerGRU sliding-window
forward -> argmax base + run-length labels, as hand-written HIP for gfx950 behind a C ABI
(`include/helen_hip.h`, built to `helen_amd/csrc/libhelen_hip.so`), plus the Python host side that
mirrors the reference's operator/loader/writer interface for this path.
"""
from .options import ImageSizeOptions, TrainOptions  # noqa: F401

__version__ = "0.1.0"
