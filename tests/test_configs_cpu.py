"""The drop-in boundary is the `target:` string of a trainer YAML, resolved by the reference's
`instantiate_from_config` (core/utils.py:37-45): `importlib.import_module(module)` + `getattr(cls)(**params)`.
The sibling configs under configs/trainer/ must resolve that way, with the YAML's own params, to modules that
keep the reference's constructor contract (core/trainer.py:26-35)."""
import importlib
import os

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def instantiate_from_config(config):
    """The resolver as the reference defines it (core/utils.py:37-45), restated."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    module, cls = config["target"].rsplit(".", 1)
    return getattr(importlib.import_module(module, package=None), cls)(**config.get("params", dict()))


def _load(name):
    with open(os.path.join(ROOT, "configs", "trainer", name)) as f:
        return yaml.safe_load(f)


def test_default_network_config_resolves_to_the_drop_in():
    cfg = _load("4x_H2_amd.yaml")["params"]
    torch.manual_seed(42)
    model = instantiate_from_config(cfg["model_config"])
    from superresolution_amd import SuperSampler
    assert type(model) is SuperSampler and model._get_name() == "SuperSampler"      # core/trainer.py:194,227
    assert cfg["model_config"]["params"].get("sr_scale", -1) == -1                 # core/trainer.py:26-29
    assert model.n_history == cfg["model_config"]["params"]["n_history"] == 1
    assert sum(p.numel() for p in model.parameters()) == 913155
    # the optimiser and scheduler entries are the reference's and build on the drop-in's parameters
    opt_cfg, sch_cfg = cfg["optimizer_config"], cfg["scheduler_config"]
    mod, cls = opt_cfg["target"].rsplit(".", 1)
    opt = getattr(importlib.import_module(mod), cls)(model.parameters(), **opt_cfg["params"])
    mod, cls = sch_cfg["target"].rsplit(".", 1)
    getattr(importlib.import_module(mod), cls)(opt, **sch_cfg["params"])
    loss = instantiate_from_config(cfg["loss_group_config"])
    assert loss.lambda_l1 == 1.0 and loss.lambda_ssim == 0.5


def test_reprfn_config_resolves_to_the_drop_in():
    cfg = _load("4x_RepRFN_amd.yaml")["params"]
    model = instantiate_from_config(cfg["model_config"])
    from superresolution_amd import RepRFN
    assert type(model) is RepRFN and model._get_name() == "RepRFN"
    assert cfg["model_config"]["params"].get("sr_scale", -1) == 4 == model.sr_scale
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) > 0
    assert len(model.state_dict()) == 354                                          # the reference's training form


def test_resolver_rejects_a_config_without_target():
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})


def test_recurrent_config_resolves_and_keeps_the_default_networks_tree():
    cfg = _load("4x_H2_recurrent_amd.yaml")["params"]
    torch.manual_seed(42)
    model = instantiate_from_config(cfg["model_config"])
    from superresolution_amd import RecurrentSuperSampler, SuperSampler
    assert type(model) is RecurrentSuperSampler and isinstance(model, SuperSampler)
    base = SuperSampler(3, 10, 1)
    # same keys; only the first G-buffer layer is wider (10 buffer + 3 fed-back channels)
    sd, sb = model.state_dict(), base.state_dict()
    assert list(sd) == list(sb)
    diff = [k for k in sd if sd[k].shape != sb[k].shape]
    assert diff == ["gbuffers_extractor.blks.0.0.weight"] and tuple(sd[diff[0]].shape) == (64, 13, 3, 3)
    with pytest.raises(NotImplementedError):
        RecurrentSuperSampler(3, 10, 2)   # the reference's trainer fills the slot for n_history == 1 only
