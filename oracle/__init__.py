"""CPU oracle for the helen polish inference path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under helen_amd/ does (tests/test_layout.py checks that).
"""
from .oracle import (  # noqa: F401
    HelenWeightsC, arbitrate, build, evaluate, gru_chunk_forward, polish_batch, polish_batch_f64, max_threads,
    set_precision, set_threads,
    weights_struct,
)
