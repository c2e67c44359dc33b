"""The model oracle against outputs of the reference module itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd.synth as synth
from oracle import model_oracle as mo


@pytest.fixture(scope="module")
def sd():
    return synth.make_state_dict("simple", seed=1234)


def test_weight_rng_is_stable(sd):
    # the fixture is only valid for exactly these weights: pin a few words of the hash stream
    assert sd["conv1.weight"].shape == (32, 1, 3, 3) and sd["lstm.weight_ih_l1"].shape == (1024, 256)
    assert sum(v.size for k, v in sd.items()) == 875394          # param count, SURVEY.md fact 4
    again = synth.make_state_dict("simple", seed=1234)
    assert all(np.array_equal(sd[k], again[k]) for k in sd)
    assert abs(float(sd["conv2.weight"].max()) - 1 / np.sqrt(288)) < 1e-4


@pytest.mark.parametrize("tag", ["32", "31"])
def test_closed_form_matches_reference_outputs(golden_simple, sd, tag):
    x = golden_simple["x" + tag]
    pooled = mo.pooled_features_np(x, sd)
    logits = mo.forward_np(x, sd)
    assert np.abs(pooled - golden_simple["pooled" + tag]).max() < 2e-5
    assert np.abs(logits - golden_simple["logits" + tag]).max() < 2e-6


def test_torch_restatement_matches_reference_outputs(golden_simple, sd):
    m = mo.torch_module_from_state_dict(sd)
    with torch.no_grad():
        y = m(torch.from_numpy(golden_simple["x32"])).numpy()
    assert np.abs(y - golden_simple["logits32"]).max() < 1e-6
    assert set(m.state_dict().keys()) == set(sd.keys())


def test_logits_do_not_depend_on_weight_hh_or_forget_gate(golden_simple, sd):
    sd2 = {k: v.copy() for k, v in sd.items()}
    for layer in range(2):
        sd2[f"lstm.weight_hh_l{layer}"][:] = 7.0
        sd2[f"lstm.weight_ih_l{layer}"][256:512] = -3.0          # forget-gate rows
        sd2[f"lstm.bias_ih_l{layer}"][256:512] = 5.0
    m = mo.torch_module_from_state_dict(sd2)
    with torch.no_grad():
        y = m(torch.from_numpy(golden_simple["x32"])).numpy()
    assert np.abs(y - golden_simple["logits32"]).max() < 1e-6


@pytest.mark.parametrize("tag", ["32", "31"])
def test_full_model_closed_form_matches_reference_class_outputs(golden_full, tag):
    # the fixture was produced by the reference's own WakewordModel class statement (ast-extracted, see make_golden.py)
    sd = synth.make_state_dict("full", seed=1234)
    x = golden_full["x" + tag]
    assert np.abs(mo.pooled_features_np(x, sd) - golden_full["pooled" + tag]).max() < 2e-5
    assert np.abs(mo.forward_np(x, sd) - golden_full["logits" + tag]).max() < 2e-6
    m = mo.torch_module_from_state_dict(sd)
    with torch.no_grad():
        y = m(torch.from_numpy(x)).numpy()
    assert np.abs(y - golden_full["logits" + tag]).max() < 1e-6


def test_full_model_closed_form_vs_torch_layers():
    # a second weight set / input range: the closed form against the same torch constructors
    sd = synth.make_state_dict("full", seed=99)
    assert sum(v.size for v in sd.values()) == 1014786           # model_architecture.txt:10
    x = synth.normal(5, 2 * 80 * 32).astype(np.float32).reshape(2, 1, 80, 32) * 20 - 40
    m = mo.torch_module_from_state_dict(sd)
    with torch.no_grad():
        y = m(torch.from_numpy(x)).numpy()
    assert np.abs(mo.forward_np(x, sd) - y).max() < 5e-6
