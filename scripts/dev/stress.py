"""Race hunt: the same 4096 + 37 windows polished many times per arithmetic mode (and through the
evaluation entry); every repetition must equal the first bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.options import TrainOptions  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    w = make_weights(seed=20260928, input_scale=1.0 / 64.0)
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 4096 + 37
    img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda", generator=g)
    lb = torch.randint(0, 5, (n, 1000), dtype=torch.uint8, device="cuda", generator=g)
    lr = torch.randint(0, 11, (n, 1000), dtype=torch.uint8, device="cuda", generator=g)
    for precision in ("fp32", "fp32x3", "bf16"):
        eng = HelenEngine(w, device=0, max_windows=4096, precision=precision)
        first = eng.polish(img, want_acc=True)
        bad = 0
        for _ in range(reps):
            again = eng.polish(img, want_acc=True)
            bad += int(not all(torch.equal(a, b) for a, b in zip(first, again)))
        cm_b = torch.zeros((5, 5), dtype=torch.int64, device="cuda")
        cm_r = torch.zeros((11, 11), dtype=torch.int64, device="cuda")
        s0 = eng.evaluate(img, lb, lr, TrainOptions.CLASS_WEIGHTS, cm_b, cm_r).clone()
        c0 = (cm_b.clone(), cm_r.clone())
        for _ in range(max(1, reps // 10)):
            cm_b.zero_()
            cm_r.zero_()
            s1 = eng.evaluate(img, lb, lr, TrainOptions.CLASS_WEIGHTS, cm_b, cm_r)
            bad += int(not (torch.equal(s0, s1) and torch.equal(c0[0], cm_b) and torch.equal(c0[1], cm_r)))
        torch.cuda.synchronize()
        print("%-7s %d repetitions, %d differing" % (precision, reps, bad), flush=True)
        eng.close()
        if bad:
            sys.exit(1)


if __name__ == "__main__":
    main()
