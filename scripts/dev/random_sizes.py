"""Developer probe / soak: helen_polish_batch at random call sizes -- the kernels the library picks by itself (quarter / half
tiles, projections in position runs, split calls, tile pairs) against the plain one-tile-per-workgroup sequence.  Labels and
accumulators must be EQUAL at every size.   python scripts/dev/random_sizes.py [n_sizes] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

PLAIN = {"HELEN_GRU_HALF8": "0", "HELEN_GRU_QUARTER4": "0", "HELEN_SPLIT": "0", "HELEN_DEC_WSP": "0",
         "HELEN_GRU_PAIR": "0", "HELEN_GRU_SINGLE8": "0", "HELEN_DEC_WS": "0"}


def main():
    n_sizes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    g = torch.Generator().manual_seed(seed)
    sizes = sorted(set([1, 15, 16, 17, 511, 512, 513, 1024, 1025, 1360, 1361, 2048, 2049, 3824, 3825, 4096] +
                       torch.randint(1, 4097, (n_sizes,), generator=g).tolist()))
    eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=0, max_windows=4096)
    gd = torch.Generator(device="cuda").manual_seed(seed)
    img = torch.randint(0, 256, (4096, 1000, 90), dtype=torch.uint8, device="cuda", generator=gd)
    img[::3] = (img[::3] > 243).to(torch.uint8) * img[::3]          # every third window pileup-like (sparse)
    bad = 0
    for n in sizes:
        for k in PLAIN:
            os.environ.pop(k, None)
        eng.reload_overrides()           # (the switches are read when an engine is created; this reads them again)
        got = [t.clone() for t in eng.polish(img[:n], want_acc=True)]
        got2 = [t.clone() for t in eng.polish(img[:n], want_acc=True)]      # the same call again
        os.environ.update(PLAIN)
        eng.reload_overrides()
        want = eng.polish(img[:n], want_acc=True)
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(want, got, got2))
        bad += not ok
        print("n=%4d (%3d tiles): %s" % (n, (n + 15) // 16, "equal" if ok else "DIFFERENT"), flush=True)
    print("%d sizes, %d different" % (len(sizes), bad))
    eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
