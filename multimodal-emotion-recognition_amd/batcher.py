"""Device-resident dialogue batcher.

The reference builds every batch on the host with three full-DataFrame scans per utterance
(src/dataset.py:35,43-45) and pads with ``apply_padding`` (src/utils.py:15-31).  Here the dialogue -> row index is
built ONCE, both embedding tables live in HBM, and a batch is one gather kernel that writes straight into the
plan's staging buffers (features, labels, key-padding mask) - no host copies in the training loop.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import runtime


def build_row_index(dialogue_ids: Sequence[int], utterance_ids: Sequence[int]) -> List[np.ndarray]:
    """Rows of each dialogue in Utterance_ID order, dialogues in order of first appearance (== Dataset.dialogue_ids)."""
    dialogue_ids = np.asarray(dialogue_ids)
    utterance_ids = np.asarray(utterance_ids)
    groups = {}
    for row, d in enumerate(dialogue_ids.tolist()):
        groups.setdefault(d, []).append(row)
    return [np.asarray(r)[np.argsort(utterance_ids[np.asarray(r)], kind="stable")] for r in groups.values()]


class DeviceDialogueBatcher:
    def __init__(self, text_table: torch.Tensor, audio_table: torch.Tensor, labels: torch.Tensor,
                 dialogue_rows: Sequence[np.ndarray], device="cuda"):
        runtime.require_gpu()
        self.text = text_table.to(device=device, dtype=torch.float32).contiguous()
        self.audio = audio_table.to(device=device, dtype=torch.float32).contiguous()
        self.labels = labels.to(device=device, dtype=torch.int64).contiguous()
        self.rows = [np.asarray(r, dtype=np.int32) for r in dialogue_rows]
        self.device = self.text.device

    def __len__(self) -> int:
        return len(self.rows)

    def slot_index(self, dialogue_ids: Sequence[int]) -> torch.Tensor:
        """int32 [B, L] table row of every token slot (-1 = pad), L = longest dialogue of the batch."""
        L = max(len(self.rows[i]) for i in dialogue_ids)
        idx = np.full((len(dialogue_ids), L), -1, dtype=np.int32)
        for b, i in enumerate(dialogue_ids):
            idx[b, : len(self.rows[i])] = self.rows[i]
        return torch.from_numpy(idx)

    def gather(self, dialogue_ids: Sequence[int], plan: Optional["runtime.Plan"] = None):
        """Fill `plan`'s staging buffers (or fresh tensors) with the batch; returns the collate_fn dict."""
        idx = self.slot_index(dialogue_ids)
        b_in, l_in = idx.shape
        if plan is not None and (plan.B, plan.L) != (b_in, l_in):
            # a bucketed plan (runtime.Plan.set_inputs): the batch sits in the top-left corner, every other slot is padding
            assert plan.B >= b_in and plan.L >= l_in, "batch does not fit the plan"
            full = torch.full((plan.B, plan.L), -1, dtype=torch.int32)
            full[:b_in, :l_in] = idx
            idx = full
        idx = idx.to(self.device, non_blocking=True)
        B, L = idx.shape
        T = B * L
        if plan is not None:
            plan.in_B, plan.in_L = b_in, l_in
            text, audio, kp, lab = plan.text_in, plan.audio_in, plan.keypad_in, plan.labels_in
        else:
            text = torch.empty(T, self.text.shape[1], device=self.device)
            audio = torch.empty(T, self.audio.shape[1], device=self.device)
            kp = torch.empty(T, dtype=torch.uint8, device=self.device)
            lab = torch.empty(T, dtype=torch.int64, device=self.device)
        runtime.check(runtime.lib().m2f_gather_dialogues(
            self.text.data_ptr(), self.text.shape[1], self.audio.data_ptr(), self.audio.shape[1], self.labels.data_ptr(),
            idx.data_ptr(), T, text.data_ptr(), text.stride(0), audio.data_ptr(), audio.stride(0), kp.data_ptr(),
            lab.data_ptr(), runtime.stream_ptr()), "m2f_gather_dialogues")
        self._keep = idx
        if plan is not None and b_in < B:
            kp.view(B, L)[b_in:, 0] = 0                # one live, unlabeled slot per filler dialogue (Plan.set_inputs)
        if plan is not None:
            return {"text": text, "audio": audio, "padding_mask": kp.view(B, L)[:b_in, :l_in].bool(),
                    "emotion": lab.view(B, L)[:b_in, :l_in]}
        return {"text": text.view(B, L, -1) if plan is None else text, "audio": audio.view(B, L, -1) if plan is None else audio,
                "padding_mask": kp.view(B, L).bool(), "emotion": lab.view(B, L)}
