"""CPU checks of the drop-in boundary: libsfhip.so loads and exports every symbol that
include/sfhip.h declares; the modules expose the reference's state_dict keys; GPU-only entry
points refuse CPU tensors loudly instead of falling back."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_pkg


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sfhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    sf = load_pkg()
    lib = sf.ops.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in sfhip.h but not exported"
    assert sorted(sf.ops.EXPORTS) == syms
    assert lib.sf_version() >= 1


def test_state_dict_keys_match_reference_layout():
    sf = load_pkg()
    cfg = {"data": {"image_size": [64, 64]}, "model": {"input_channels": 3, "hidden_dim": 16,
                                                       "temporal_module": "Transformer"}}
    sd = sf.StreamflowPredictionModel(cfg).state_dict()
    for k, shape in {
        "cnn_encoder.encoder.0.weight": (16, 3, 3, 3),
        "cnn_encoder.encoder.1.running_var": (16,),
        "cnn_encoder.encoder.4.weight": (32, 16, 3, 3),
        "cnn_encoder.encoder.5.num_batches_tracked": (),
        "temporal_module.input_proj.weight": (16, 32 * 32 * 32),
        "temporal_module.pos_encoder.pe": (1, 100, 16),
        "temporal_module.transformer_encoder.layers.1.self_attn.in_proj_weight": (48, 16),
        "temporal_module.transformer_encoder.layers.0.linear1.weight": (64, 16),
        "temporal_module.transformer_encoder.layers.0.norm2.bias": (16,),
        "temporal_to_spatial.weight": (32 * 32 * 32, 16),
        "decoder.conv1.weight": (64, 32, 3, 3),
        "decoder.output_conv.weight": (1, 64, 1, 1),
    }.items():
        assert tuple(sd[k].shape) == shape, k
    cfg["model"]["temporal_module"] = "LSTM"
    sd = sf.StreamflowPredictionModel(cfg).state_dict()
    assert tuple(sd["temporal_module.lstm.weight_hh_l1"].shape) == (64, 16)
    dsd = sf.PatchDiscriminator(1, 64).state_dict()
    assert tuple(dsd["model.0.weight"].shape) == (64, 1, 4, 4)
    assert tuple(dsd["model.4.0.weight"].shape) == (512, 256, 4, 4)
    assert tuple(dsd["model.3.1.running_mean"].shape) == (256,)
    assert tuple(dsd["model.5.weight"].shape) == (1, 512, 4, 4)


def test_no_cpu_fallback():
    sf = load_pkg()
    enc = sf.CNNEncoder(4, 16, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.zeros(1, 4, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        sf.ImageLoss()(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
