"""Dialogue-level dataset with the reference's surface (``src/dataset.py``): ``Dataset(mode)``, ``__getitem__``
-> {"text": [n, d_t], "audio": [n, d_a], "emotion": list of [1] tensors}, ``collate_fn`` -> {"text": [B, L, d_t],
"audio": [B, L, d_a], "padding_mask": bool [B, L], "emotion": int64 [B, L] (-1 on pads)}.

Unlike the reference (three full-DataFrame scans per utterance, src/dataset.py:35,43-45) the dialogue -> row
index is built ONCE at construction, so ``__getitem__`` is two tensor gathers."""
import os
import pickle

import numpy as np
import torch

from utils import get_config, get_text

EMOTIONS = {"neutral": 0, "joy": 1, "sadness": 2, "anger": 3, "surprise": 4, "fear": 5, "disgust": 6}


def build_dialogue_index(dialogue_ids, utterance_ids):
    """Rows of each dialogue in Utterance_ID order; dialogues in order of first appearance."""
    dialogue_ids = np.asarray(dialogue_ids)
    utterance_ids = np.asarray(utterance_ids)
    order = {}
    for row, d in enumerate(dialogue_ids.tolist()):
        order.setdefault(d, []).append(row)
    index = []
    for d, rows in order.items():
        rows = np.asarray(rows)
        index.append(rows[np.argsort(utterance_ids[rows], kind="stable")])
    return list(order.keys()), index


def tokenised_contexts(table, tokenizer, max_tokens=64):
    """Token ids of every utterance WITH ITS CONTEXT, the input of the in-loop text encoder (BASELINE C5; `runtime.text_encoder`):
    the strings of the reference's text feature extractor (src/feature_extractors/text/dataset.py:28-34 -> utils.py:61-92,
    restated in mer_amd.text_context), tokenised by the CALLER's tokenizer the way that dataset does (:36-44: padding to
    max_length, truncation) -> (int64 ids [N, max_tokens], int64 mask [N, max_tokens]) in table row order."""
    from mer_amd.text_context import build_contexts
    texts = build_contexts(table["Utterance"].tolist(), table["Dialogue_ID"].tolist(), table["Utterance_ID"].tolist(), tokenizer.sep_token)
    enc = tokenizer(texts, padding="max_length", truncation=True, max_length=int(max_tokens), return_tensors="pt")
    return enc["input_ids"].to(torch.int64), enc["attention_mask"].to(torch.int64)


class Dataset(torch.utils.data.Dataset):
    def __init__(self, mode="train", text_embeddings=None, audio_embeddings=None, table=None, token_ids=None, token_mask=None):
        """token_ids / token_mask ([N, S] int64, rows = table rows; `tokenised_contexts`): items and batches ALSO carry "text_ids" /
        "text_mask" - what the in-loop text encoder consumes instead of the pre-extracted "text" rows (`runtime.text_encoder`)."""
        super().__init__()
        self.mode = mode
        self.token_ids, self.token_mask = token_ids, token_mask
        if text_embeddings is None or audio_embeddings is None:
            config = get_config()
            with open(os.path.join(os.path.abspath(config.embeddings.text), f"{mode}.pkl"), "rb") as f:
                text_embeddings = pickle.load(f)
            with open(os.path.join(os.path.abspath(config.embeddings.audio), f"{mode}.pkl"), "rb") as f:
                audio_embeddings = pickle.load(f)
        self.text_embeddings, self.audio_embeddings = text_embeddings, audio_embeddings
        self.text = get_text(mode) if table is None else table
        self.text["Emotion"] = self.text["Emotion"].map(lambda e: EMOTIONS.get(e, e))
        self.dialogue_ids, self.rows = build_dialogue_index(self.text["Dialogue_ID"].to_numpy(),
                                                            self.text["Utterance_ID"].to_numpy())
        self._labels = torch.as_tensor(self.text["Emotion"].to_numpy().astype(np.int64))
        print(f"Loaded {len(self.dialogue_ids)} dialogues for {self.mode}ing")

    def __len__(self):
        return len(self.dialogue_ids)

    def __getitem__(self, idx):
        rows = torch.as_tensor(self.rows[idx])
        item = {"text": self.text_embeddings[rows], "audio": self.audio_embeddings[rows],
                "emotion": [e.reshape(1) for e in self._labels[rows]]}
        if self.token_ids is not None:
            item["text_ids"] = self.token_ids[rows]
            item["text_mask"] = self.token_mask[rows] if self.token_mask is not None else torch.ones_like(item["text_ids"])
        return item

    def get_labels(self):
        return self.text["Emotion"].to_numpy()


def collate_fn(batch):
    B = len(batch)
    L = max(d["text"].shape[0] for d in batch)
    text = batch[0]["text"].new_zeros(B, L, batch[0]["text"].shape[1])
    audio = batch[0]["audio"].new_zeros(B, L, batch[0]["audio"].shape[1])
    emotion = torch.full((B, L), -1, dtype=torch.int64)          # -1 = ignored by the criterion
    for i, d in enumerate(batch):
        n = d["text"].shape[0]
        text[i, :n], audio[i, :n] = d["text"], d["audio"]
        emotion[i, :n] = torch.cat([torch.as_tensor(e).reshape(1) for e in d["emotion"]]).to(torch.int64)
    out = {"text": text, "audio": audio, "padding_mask": emotion == -1, "emotion": emotion}
    if "text_ids" in batch[0]:                                     # in-loop text encoder: [B, L, S] token ids + attention mask, zero on pads
        S = batch[0]["text_ids"].shape[1]
        ids = torch.zeros(B, L, S, dtype=torch.int64)
        msk = torch.zeros(B, L, S, dtype=torch.int64)
        for i, d in enumerate(batch):
            n = d["text_ids"].shape[0]
            ids[i, :n], msk[i, :n] = d["text_ids"], d["text_mask"]
        out["text_ids"], out["text_mask"] = ids, msk
    return out


class DeviceLoader:
    """Drop-in for ``torch.utils.data.DataLoader(Dataset, collate_fn=collate_fn, batch_size, shuffle)`` with both embedding
    tables resident in HBM (``runtime.device_batcher: True``): one gather kernel per batch fills the same
    {"text", "audio", "padding_mask", "emotion"} dict, on the device, with dialogues in DataLoader order (a fresh
    ``torch.randperm`` per epoch when shuffling, the last batch partial - torch's ``drop_last=False`` default)."""

    def __init__(self, dataset, batch_size=32, shuffle=False, device="cuda", seed=None, **_ignored_loader_kwargs):
        from mer_amd.batcher import DeviceDialogueBatcher
        self.batch_size, self.shuffle = int(batch_size), bool(shuffle)
        self.n = len(dataset)
        self.batcher = DeviceDialogueBatcher(dataset.text_embeddings, dataset.audio_embeddings, dataset._labels, dataset.rows,
                                             device=device)
        self.generator = torch.Generator()
        if seed is not None:
            self.generator.manual_seed(int(seed))

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.generator).tolist() if self.shuffle else list(range(self.n))
        for start in range(0, self.n, self.batch_size):
            yield self.batcher.gather(order[start: start + self.batch_size])
