#!/usr/bin/env python3
"""Developer probe: (1) helen_polish_host = host uint8 images -> labels in host memory (PCIe
included, pinned double buffering); (2) the whole call_consensus path from an HDF5 image directory
to the prediction HDF5 (reader workers + device + writer thread)."""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# torch and the engine are imported inside the probes: reader / writer processes re-import this file
# (spawn) and must stay as light as they are under `python -m helen_amd`


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=16384)
    ap.add_argument("--h5-windows", type=int, default=8192)
    ap.add_argument("--workers", type=str, default="8", help="reader processes; comma list = one run each")
    ap.add_argument("--writers", type=str, default="1", help="$HELEN_WRITERS; comma list = one run each")
    ap.add_argument("--skip-host", action="store_true")
    args = ap.parse_args()

    from helen_amd.weights import make_weights
    w = make_weights(input_scale=1.0 / 64.0)
    if not args.skip_host:
        host_probe(w, args)
    e2e_probe(w, args)


def host_probe(w, args):
    import torch
    from helen_amd.engine import HelenEngine
    eng = HelenEngine(w, device=0, max_windows=4096)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(args.windows, 1000, 90), dtype=np.uint8)
    eng.polish_host(img[:4096])
    t0 = time.time()
    b, r = eng.polish_host(img)
    dt = time.time() - t0
    print("polish_host: %d windows in %.3f s = %.0f windows/s (H2D %.2f GB/s incl.)"
          % (args.windows, dt, args.windows / dt, img.nbytes / dt / 1e9))
    dev = torch.from_numpy(img[:4096]).cuda()
    bd, rd = eng.polish(dev)
    assert np.array_equal(bd.cpu().numpy(), b[:4096])
    eng.close()


def e2e_probe(w, args):
    from helen_amd.call_consensus import call_consensus
    from helen_amd.model_handler import ModelHandler
    from helen_amd.synthetic import write_image_dir
    d = tempfile.mkdtemp(prefix="helen_e2e_")
    try:
        model = os.path.join(d, "m.pkl")
        ModelHandler.save_model(w, None, 128, 1, 0, model)
        t0 = time.time()
        write_image_dir(os.path.join(d, "img"), args.h5_windows, n_files=16, direct=True)
        print("wrote %d windows of synthetic HDF5 in %.1f s" % (args.h5_windows, time.time() - t0))
        for workers in [int(x) for x in args.workers.split(",")]:
            for writers in [int(x) for x in args.writers.split(",")]:
                os.environ["HELEN_WRITERS"] = str(writers)
                out = os.path.join(d, "out_%d_%d" % (workers, writers))
                t0 = time.time()
                call_consensus(os.path.join(d, "img"), model, 256, workers, 1, out, "p", True, "0", 1)
                dt = time.time() - t0
                print("call_consensus end-to-end (HDF5 in -> HDF5 out, %d reader workers, %d writers): "
                      "%d windows in %.2f s = %.0f windows/s"
                      % (workers, writers, args.h5_windows, dt, args.h5_windows / dt), flush=True)
                shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":   # worker processes re-import this file (spawn): keep it inert
    main()
