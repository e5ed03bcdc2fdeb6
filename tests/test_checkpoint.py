"""helen_amd/checkpoint.py reads the reference's `.pkl` (a torch.save'd dict, models/ModelHander.py:109-133) WITHOUT torch:
it must give what torch.load gives, for both container formats torch has written over the years."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _checkpoint(tmp_path, legacy, module_prefix=False, with_optimizer=True):
    import torch

    from helen_amd.weights import make_weights
    w = make_weights(seed=11)
    sd = {("module." + k if module_prefix else k): torch.from_numpy(v.copy()) for k, v in w.items()}
    # a transposed (non-contiguous) and a sliced (offset) parameter, as a checkpoint of a live model can hold them
    k0 = next(iter(sd))
    sd[k0] = sd[k0].t().contiguous().t()
    big = torch.arange(4000, dtype=torch.float32)
    k1 = [k for k in sd if k.endswith("dense1_base.bias")][0]
    sd[k1] = big[100:105]
    w[k1[7:] if module_prefix else k1] = big[100:105].numpy()
    opt = {}
    if with_optimizer:
        opt = {"state": {0: {"step": 7, "exp_avg": torch.ones(3, 2), "exp_avg_sq": torch.zeros(3, 2, dtype=torch.float64)}},
               "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.999), "params": [0]}]}
    path = str(tmp_path / ("legacy.pkl" if legacy else "zip.pkl"))
    torch.save({"model_state_dict": sd, "model_optimizer": opt, "hidden_size": 128, "gru_layers": 1, "epochs": 3}, path,
               _use_new_zipfile_serialization=not legacy)
    return path, w


@pytest.mark.parametrize("legacy", [False, True])
@pytest.mark.parametrize("module_prefix", [False, True])
def test_checkpoint_reader_equals_torch_load(tmp_path, legacy, module_prefix):
    import torch

    from helen_amd import checkpoint
    path, w = _checkpoint(tmp_path, legacy, module_prefix)
    state, hidden, layers, epochs = checkpoint.load_simple_model_state(path)
    assert (hidden, layers, epochs) == (128, 1, 3)
    ref = torch.load(path, map_location="cpu", weights_only=False)
    assert list(state) == [k[7:] if k.startswith("module.") else k for k in ref["model_state_dict"]]
    for k, v in ref["model_state_dict"].items():
        mine = state[k[7:] if k.startswith("module.") else k]
        assert mine.dtype == np.float32 and mine.flags.c_contiguous and np.array_equal(mine, v.numpy()), k
    full = checkpoint.load(path)
    assert full["model_optimizer"]["state"][0]["step"] == 7
    assert np.array_equal(full["model_optimizer"]["state"][0]["exp_avg_sq"], np.zeros((3, 2)))
    assert full["model_optimizer"]["state"][0]["exp_avg_sq"].dtype == np.float64


def test_checkpoint_reader_refuses_what_it_does_not_know(tmp_path):
    """Anything but tensors, storages and plain containers is refused (the caller then uses torch.load): no arbitrary
    class of a pickle is ever instantiated by this reader."""
    import pickle

    import torch

    from helen_amd import checkpoint
    p = str(tmp_path / "odd.pkl")
    torch.save({"model_state_dict": {"w": torch.ones(2)}, "model_optimizer": torch.optim.SGD, "hidden_size": 128,
                "gru_layers": 1, "epochs": 0}, p)
    with pytest.raises(checkpoint.UnsupportedCheckpoint):
        checkpoint.load(p)
    q = str(tmp_path / "plain.pkl")
    with open(q, "wb") as f:
        pickle.dump({"a": 1}, f)
    with pytest.raises(checkpoint.UnsupportedCheckpoint):
        checkpoint.load(q)
    r = str(tmp_path / "cut.pkl")
    path, _ = _checkpoint(tmp_path, True)
    with open(path, "rb") as f, open(r, "wb") as g:
        g.write(f.read()[:-1000])
    with pytest.raises(checkpoint.UnsupportedCheckpoint):
        checkpoint.load(r)


def test_tensor_geometry_is_checked_against_the_storage():
    """A pickled (offset, size, stride) that does not fit its storage -- a truncated or corrupt file -- is refused instead of
    becoming an out-of-bounds strided view."""
    from helen_amd import checkpoint
    storage = np.arange(12, dtype=np.float32)
    ok = checkpoint._rebuild_tensor(storage, 2, (2, 5), (5, 1))
    assert ok.shape == (2, 5) and ok[0, 0] == 2 and ok[1, 4] == 11
    assert checkpoint._rebuild_tensor(storage, 0, (3, 4), (1, 3)).tolist() == storage.reshape(4, 3).T.tolist()
    for offset, size, stride in ((3, (2, 5), (5, 1)), (0, (13,), (1,)), (0, (2, 2), (100, 1)), (-1, (2,), (1,)), (0, (2,), (-1,)),
                                 (0, (2, 2), (1,)), (12, (), ())):
        with pytest.raises(checkpoint.UnsupportedCheckpoint):
            checkpoint._rebuild_tensor(storage, offset, size, stride)


def test_checkpoint_reader_does_not_import_torch(tmp_path):
    path, _ = _checkpoint(tmp_path, False)
    code = ("import sys; sys.path.insert(0, %r); from helen_amd import checkpoint; s = checkpoint.load_simple_model_state(%r); "
            "assert 'torch' not in sys.modules, 'torch was imported'; print(len(s[0]))" % (ROOT, path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "20", out.stderr[-2000:]
