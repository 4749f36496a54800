"""Config / padding / transcript helpers with the reference's names (``src/utils.py``): ``get_config``,
``apply_padding``, ``get_text``.  ``munch`` is not required: ``AttrDict`` gives the same attribute access."""
import os

import torch
import yaml


class AttrDict(dict):
    """dict with attribute access, recursively (Munch-like); still a dict, so ``DataLoader(**cfg)`` works."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(x) for x in v]
        return v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = self._wrap(value)


config = None
CONFIG_PATH = "./src/config.yaml"      # same relative path as the reference: launch from the repo root


def get_config(path=None):
    """Process-global config singleton (reference src/utils.py:8-13)."""
    global config
    if config is None or path is not None:
        with open(path or CONFIG_PATH, "rt", encoding="utf-8") as f:
            config = AttrDict(yaml.safe_load(f))
    return config


def apply_padding(tensors_list, padding_value=0):
    """[1, n_i, ...] tensors -> ([B, max_n, ...] padded with `padding_value`, lengths int64 [B])."""
    lengths = torch.tensor([t.shape[1] for t in tensors_list], dtype=torch.int64)
    longest = int(lengths.max())
    first = tensors_list[0]
    out = first.new_full((len(tensors_list), longest) + tuple(first.shape[2:]), padding_value)
    for i, t in enumerate(tensors_list):
        out[i, : t.shape[1]] = t[0]
    return out, lengths


_CORRUPTED = {"train": [(125, 3)], "val": [(110, 7)], "test": [(38, 4), (220, 0)]}   # clips dropped upstream
_SPLIT_FILE = {"train": "train_sent_emo.csv", "val": "dev_sent_emo.csv", "test": "test_sent_emo.csv"}
_CP1252 = {"\x85": "…", "\x91": "‘", "\x92": "’", "\x93": "“", "\x94": "”",
           "\x96": "–", "\x97": "—", "\xa0": " "}


def get_text(mode="train"):
    """MELD transcript table of a split: columns Utterance, Emotion, Dialogue_ID, Utterance_ID; the rows of the
    corrupted clips removed and the index reset, so row i <-> row i of the embedding pickles."""
    import pandas as pd
    if mode not in _SPLIT_FILE:
        raise ValueError(f"Invalid mode {mode}")
    path = os.path.join(os.path.abspath("data"), "MELD.Raw", _SPLIT_FILE[mode])
    if not os.path.exists(path):
        raise ValueError(f"Dataset not found at {path}")
    df = pd.read_csv(path, usecols=["Utterance", "Emotion", "Dialogue_ID", "Utterance_ID"])
    for dia, utt in _CORRUPTED[mode]:
        df = df[(df["Dialogue_ID"] != dia) | (df["Utterance_ID"] != utt)]
    df = df.reset_index(drop=True)
    table = str.maketrans(_CP1252)
    df["Utterance"] = df["Utterance"].map(lambda s: s.translate(table))
    return df
