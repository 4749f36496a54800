"""Context construction of the in-loop text feature extractor (BASELINE config C5).

Restates /root/reference/src/feature_extractors/text/utils.py:61-92 (`get_utterance_with_context`, called per item by
text/dataset.py:28-34 with the tokenizer's separator `</s>`): an utterance is encoded together with the previous and the next
utterance OF ITS DIALOGUE (neighbours in sorted Utterance_ID order), joined as

    "{prev} {sep} {utt} {sep} {next}"        with an absent neighbour left out:  "{sep} {utt} ..."  /  "... {utt} {sep}"

The reference scans the whole DataFrame three times per utterance (O(N) each, O(N^2) per epoch); `build_contexts` indexes the
dialogues once and produces every string in one pass - on the host, it is string work; the encoder that consumes the
tokenised strings is mer_amd.roberta (HIP).  `cls_dump` is the protocol of text/embeddings.py:64-90: one [CLS] row per
dataset index, float32 [N, d], pickled to embeddings/<name>/<mode>.pkl - the file format src/dataset.py:14-17 reads back.
"""
from __future__ import annotations

import os
import pickle
from typing import Iterable, List, Sequence

import torch


def utterance_with_context(utterances: Sequence[str], dialogue_ids: Sequence[int], utterance_ids: Sequence[int], idx: int,
                           separator: str) -> str:
    """One row, the reference's way (text/utils.py:61-92).  Raises ValueError as the reference does when the row's Utterance_ID
    is not found among its dialogue's ids (cannot happen for a consistent table)."""
    dia, uid = int(dialogue_ids[idx]), int(utterance_ids[idx])
    rows = [i for i in range(len(utterances)) if int(dialogue_ids[i]) == dia]
    ids = sorted(int(utterance_ids[i]) for i in rows)
    if uid not in ids:
        raise ValueError(f"Utterance ID {uid} not found in dialogue ID {dia}")
    pos = ids.index(uid)
    first_row_of = {}
    for i in rows:                                            # `.iloc[0]` of the reference: the first row carrying that id
        first_row_of.setdefault(int(utterance_ids[i]), i)
    text = utterances[idx]
    text = f"{utterances[first_row_of[ids[pos - 1]]]} {separator} {text}" if pos > 0 else f"{separator} {text}"
    text = f"{text} {separator} {utterances[first_row_of[ids[pos + 1]]]}" if pos < len(ids) - 1 else f"{text} {separator}"
    return text


def build_contexts(utterances: Sequence[str], dialogue_ids: Sequence[int], utterance_ids: Sequence[int], separator: str) -> List[str]:
    """Every row's context string in one pass (same strings as `utterance_with_context` row by row)."""
    n = len(utterances)
    by_dia = {}
    for i in range(n):
        by_dia.setdefault(int(dialogue_ids[i]), []).append(i)
    out: List[str] = [""] * n
    for rows in by_dia.values():
        first_row_of = {}
        for i in rows:
            first_row_of.setdefault(int(utterance_ids[i]), i)
        ids = sorted(int(utterance_ids[i]) for i in rows)    # duplicates stay duplicates, as in the reference's sorted list
        for i in rows:
            pos = ids.index(int(utterance_ids[i]))
            text = utterances[i]
            text = f"{utterances[first_row_of[ids[pos - 1]]]} {separator} {text}" if pos > 0 else f"{separator} {text}"
            text = f"{text} {separator} {utterances[first_row_of[ids[pos + 1]]]}" if pos < len(ids) - 1 else f"{text} {separator}"
            out[i] = text
    return out


def cls_dump(encoder, batches: Iterable[dict], n_items: int, path: str, mode: str) -> torch.Tensor:
    """text/embeddings.py:64-90: `batches` yield {"idx": LongTensor[b], "text": token ids [b, S], "attention_mask": [b, S]};
    row idx[j] of the result is the [CLS] (position 0) hidden state of item j; the tensor is pickled to <path>/<mode>.pkl."""
    out = None
    with torch.inference_mode():
        for batch in batches:
            cls = encoder.cls_embeddings(batch["text"], batch["attention_mask"]).float().cpu()
            if out is None:
                out = torch.zeros(n_items, cls.shape[1], dtype=torch.float32)
            out[batch["idx"].cpu()] = cls
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(os.path.abspath(path), f"{mode}.pkl"), "wb") as f:
        pickle.dump(out, f)
    return out
