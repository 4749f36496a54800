"""Writes tests/golden/roberta_*.npz by running the REAL third-party model the reference uses for its text embeddings,
transformers.RobertaModel(add_pooling_layer=False) (reference src/feature_extractors/text/model.py:16), in eval mode on
CPU, with the deterministic weights / token batches of synth_roberta.py.  Pretrained weights cannot be fetched offline, so
the pin is "same architecture code, synthetic weights".  Fixtures hold outputs only (CLS rows + a few hidden rows).
usage: python tests/golden/make_golden_roberta.py"""
import os
import sys

import numpy as np
import torch
from transformers import RobertaConfig, RobertaModel

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth_roberta as SR  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(8)
    for name, (c, B, S, lengths) in SR.CASES.items():
        model = RobertaModel(RobertaConfig(**c), add_pooling_layer=False).eval()
        sd = SR.make_state_dict(c)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
        ids, mask = SR.make_batch(c, B, S, lengths)
        with torch.inference_mode():
            hid = model(input_ids=ids, attention_mask=mask).last_hidden_state
        valid = mask.bool()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), cls=hid[:, 0, :].numpy(),
                            hidden_valid_mean=np.array([float(hid[valid].double().mean())]),
                            hidden_valid_abs=np.array([float(hid[valid].double().abs().mean())]),
                            hidden_rows=hid[0, : min(S, 8)].numpy(), hidden_last_valid=np.stack([hid[b, n - 1].numpy() for b, n in enumerate(lengths)]))
        print(name, tuple(hid.shape), float(hid[valid].abs().mean()))


if __name__ == "__main__":
    main()
