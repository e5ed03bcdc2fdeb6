"""Structural constants of the `helen polish` inference path.

Mirrors the class constants of the reference (`helen/modules/python/Options.py:13-29`):
image geometry (90 features x 1000 positions), the sliding window (100 wide, jump 50),
the GRU width (128, one layer, bidirectional) and the two label alphabets (5 bases, 11 run
lengths).  Everything in this package and in the HIP library is sized from these.
"""


class ImageSizeOptions(object):
    IMAGE_HEIGHT = 90          # features per pileup position (Options.py:14)
    IMAGE_CHANNELS = 1
    SEQ_LENGTH = 1000          # positions per window (Options.py:16)
    SEQ_OVERLAP = 200
    LABEL_LENGTH = SEQ_LENGTH
    TOTAL_BASE_LABELS = 5      # '', A, C, G, T
    TOTAL_RLE_LABELS = 11      # run lengths 0..10


class TrainOptions(object):
    TRAIN_WINDOW = 100         # chunk width fed to the GRU (Options.py:25)
    WINDOW_JUMP = 50           # chunk stride (Options.py:26)
    GRU_LAYERS = 1
    HIDDEN_SIZE = 128
    # run-length class weights of the evaluation / training loss (Options.py:29)
    CLASS_WEIGHTS = [0.3, 0.5, 0.5, 0.5, 0.5, 0.8, 0.9, 1.0, 1.0, 1.0, 0.9]


def chunk_starts(seq_length=ImageSizeOptions.SEQ_LENGTH,
                 window=TrainOptions.TRAIN_WINDOW,
                 jump=TrainOptions.WINDOW_JUMP):
    """Chunk start offsets exactly as the reference loop produces them
    (`models/predict_gpu.py:114-117`): 0, 50, ..., 900 -> 19 chunks."""
    out = []
    for i in range(0, seq_length, jump):
        if i + window > seq_length:
            break
        out.append(i)
    return out
