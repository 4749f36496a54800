"""CPU: the text-encoder oracle (oracle/roberta_oracle.py) against the fixtures written by the real
transformers.RobertaModel (tests/golden/make_golden_roberta.py)."""
import os

import numpy as np
import pytest
import torch

import synth_roberta as SR
from oracle import roberta_oracle as RO


@pytest.mark.parametrize("name", list(SR.CASES))
def test_roberta_oracle_matches_transformers_fixture(golden_dir, name):
    c, B, S, lengths = SR.CASES[name]
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = SR.make_state_dict(c)
    ids, mask = SR.make_batch(c, B, S, lengths)
    hid = RO.forward(sd, c, ids, mask)
    assert np.abs(hid[:, 0, :].numpy() - fx["cls"]).max() < 2e-5
    assert np.abs(hid[0, : min(S, 8)].numpy() - fx["hidden_rows"]).max() < 2e-5
    last = np.stack([hid[b, n - 1].numpy() for b, n in enumerate(lengths)])
    assert np.abs(last - fx["hidden_last_valid"]).max() < 2e-5
    valid = mask.bool()
    assert abs(float(hid[valid].double().abs().mean()) - float(fx["hidden_valid_abs"][0])) < 1e-5


def test_position_ids_follow_transformers_rule():
    ids = torch.tensor([[0, 5, 6, 2, 1, 1], [0, 9, 2, 1, 1, 1]])
    assert RO.position_ids(ids, 1).tolist() == [[2, 3, 4, 5, 1, 1], [2, 3, 4, 1, 1, 1]]


def test_product_module_has_transformers_state_dict_keys():
    """RobertaEncoder.load_state_dict accepts RobertaModel.state_dict() as is (same keys, same shapes)."""
    import mer_amd  # noqa: F401
    from mer_amd.roberta import RobertaEncoder
    c = SR.CASES["roberta_tiny"][0]
    m = RobertaEncoder(c)
    want = dict(SR.state_dict_shapes(c))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    m.load_state_dict(SR.make_state_dict(c))
    import torch
    from mer_amd import runtime
    import pytest
    with pytest.raises(runtime.HipError):
        m(torch.zeros(1, 4, dtype=torch.int64))                      # no GPU here: fails loudly, no CPU fallback


def test_text_context_construction_matches_reference(golden_dir):
    """f4 / BASELINE C5: `prev </s> utterance </s> next` of src/feature_extractors/text/utils.py:61-92 - fixture written by the
    reference's own function (tests/golden/make_golden_context.py); the row-by-row restatement and the one-pass builder both
    reproduce every string (first / last / only utterance of a dialogue, gaps in the ids, rows out of order)."""
    import json
    import os
    from mer_amd import text_context as tc
    with open(os.path.join(golden_dir, "text_contexts.json"), encoding="utf-8") as f:
        fx = json.load(f)
    u, d, i, sep = fx["utterances"], fx["dialogue_ids"], fx["utterance_ids"], fx["separator"]
    assert len(u) >= 10 and any(c.startswith(sep + " ") for c in fx["contexts"]) and any(c.endswith(" " + sep) for c in fx["contexts"])
    assert [tc.utterance_with_context(u, d, i, k, sep) for k in range(len(u))] == fx["contexts"]
    assert tc.build_contexts(u, d, i, sep) == fx["contexts"]
