"""Checkpoint files: the reference's format (a bare generator state_dict, src/train.py:472-490,
read back by src/train.py:49-51 and src/evaluate.py:104-107) and the resume superset."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import close, golden, load_pkg
from test_gpu_parity import load_gold, make_batches, train_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return load_pkg()


def _trainer(sf, **kw):
    training = importlib.import_module(sf.__name__ + ".training")
    tr = training.Trainer(train_cfg("Transformer", True, 2), device="cuda", **kw)
    return tr


def _steps(tr, batches):
    out = []
    for b in batches:
        o = tr.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])
        out.append(float(o["total"].detach()))
    return out


def test_reference_format_model_file(sf, tmp_path):
    tr = _trainer(sf)
    load_gold(tr.model, "trn_g_trgan")
    _steps(tr, make_batches(2, 2, 3, 64))
    path = str(tmp_path / "best_model.pt")
    tr.save_model(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    ck = golden("checkpoint_keys")                                      # what the reference's own file holds
    assert list(sd.keys()) == [str(k) for k in ck["tr_keys"]]
    assert [list(v.shape) + [-1] * (5 - v.dim()) for v in sd.values()] == ck["tr_shapes"].tolist()
    assert [str(v.dtype) for v in sd.values()] == [str(d) for d in ck["tr_dtypes"]]
    assert all(v.dtype in (torch.float32, torch.int64) for v in sd.values())
    n_bytes = sum(v.numel() * v.element_size() for v in sd.values())
    assert os.path.getsize(path) < 1.2 * n_bytes + 65536                # views are not saved with their flat buffer
    for k, v in tr.model.state_dict().items():
        assert torch.equal(sd[k], v.cpu()), k
    tr2 = _trainer(sf, model_path=path)                                 # Trainer(config, model_path)
    assert torch.equal(tr2.opt_g.flat, tr.opt_g.flat)
    assert tr2.load_checkpoint(path) == "model"
    ev = sf.Evaluator(train_cfg("Transformer", True, 2), path)          # Evaluator(config, model_path)
    for k, v in ev.model.state_dict().items():
        assert torch.equal(sd[k], v.cpu()), k


def test_resume_continues_the_run(sf, tmp_path):
    """Two steps, save, three more steps == load into a fresh trainer, the same three steps:
    parameters of both networks, Adam moments and step counts, the discriminator schedule
    (batch counter) and the dropout seeds (CPU generator state) all carry over.  Dropout is on."""
    batches = make_batches(5, 2, 3, 64)
    torch.manual_seed(5)
    tr = _trainer(sf)
    load_gold(tr.model, "trn_g_trgan")
    load_gold(tr.discriminator, "trn_d_trgan")
    _steps(tr, batches[:2])
    path = str(tmp_path / "resume.pt")
    tr.save_checkpoint(path)
    la = _steps(tr, batches[2:])
    torch.manual_seed(99)                                               # the file's generator state must win
    tr2 = _trainer(sf)
    assert tr2.load_checkpoint(path) == "resume"
    assert tr2.batch_idx == 2 and tr2.opt_g.t == 1          # accumulation 2: one optimiser step so far
    lb = _steps(tr2, batches[2:])
    np.testing.assert_allclose(lb, la, rtol=5e-4)
    assert tr2.batch_idx == tr.batch_idx and tr2.opt_g.t == tr.opt_g.t and tr2.opt_d.t == tr.opt_d.t
    close(tr2.opt_g.flat, tr.opt_g.flat.cpu().numpy(), 2e-3)
    close(tr2.opt_d.flat, tr.opt_d.flat.cpu().numpy(), 2e-3)
    close(tr2.opt_g.m, tr.opt_g.m.cpu().numpy(), 2e-3)
    for (k, a), b in zip(tr.model.state_dict().items(), tr2.model.state_dict().values()):
        if "running" in k:
            close(b, a.cpu().numpy(), 1e-4)
