import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine
from helen_amd.weights import make_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = HelenEngine(make_weights(input_scale=1/64.), device=0, max_windows=n, precision="fp32x3")
x = torch.zeros(n, 100, 90, device="cuda"); h = torch.zeros(n, 2, 128, device="cuda")
eng.chunk_forward(x, h); torch.cuda.synchronize()
