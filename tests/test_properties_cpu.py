"""CPU property tests (hypothesis) of the host-side logic around the hot path: dialogue indexing, data-parallel
sharding and bucket geometry, flat parameter layout."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import synth
import mer_amd  # noqa: F401
from mer_amd import dp, layout
from mer_amd.batcher import build_row_index


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 12), st.integers(0, 30)), min_size=1, max_size=80, unique=True))
def test_row_index_groups_by_dialogue_in_first_appearance_order_and_sorts_utterances(pairs):
    dia = [p[0] for p in pairs]
    utt = [p[1] for p in pairs]
    rows = build_row_index(dia, utt)
    order = list(dict.fromkeys(dia))                       # dialogues in order of first appearance
    assert len(rows) == len(order)
    seen = []
    for d, r in zip(order, rows):
        r = r.tolist()
        assert all(dia[i] == d for i in r)
        assert [utt[i] for i in r] == sorted(utt[i] for i in r)
        seen += r
    assert sorted(seen) == list(range(len(pairs)))         # every table row exactly once


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 300), st.integers(1, 16))
def test_shard_dialogues_is_a_partition(n, world):
    parts = [dp.shard_dialogues(n, r, world) for r in range(world)]
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(n))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 40).map(lambda k: 64 * k), st.integers(1, 12), st.sampled_from(["fp32", "bf16"]))
def test_gradient_buckets_tile_the_buffer(n_params, n_buckets, exchange):
    buf = torch.zeros(n_params + dp.TAIL)
    red = dp.GradReducer(buf, n_params, n_buckets=n_buckets, exchange=exchange)
    for chunks, end in ((red.chunks, n_params + dp.TAIL), (red.param_chunks, n_params)):
        assert chunks[0][0] == 0 and chunks[-1][1] == end
        assert all(a < b and a % 64 == 0 for a, b in chunks)
        assert all(chunks[i][1] == chunks[i + 1][0] for i in range(len(chunks) - 1))
    assert (red.buf16 is not None) == (exchange == "bf16")
    assert red.tail.numel() == 3


@settings(max_examples=25, deadline=None)
@given(st.sampled_from([16, 24, 40, 64]), st.sampled_from([16, 32, 48]), st.sampled_from([16, 32, 64]),
       st.integers(1, 3), st.integers(1, 3), st.integers(1, 3), st.integers(1, 2))
def test_flat_layout_is_aligned_dense_and_alias_consistent(d_a, d_t, d_f, nl_a, nl_t, nl_f, n_tr):
    cfg = synth._cfg(d_a, d_t, d_f, 4, 4, 4, nl_a, nl_t, nl_f, nt_a=n_tr, nt_t=n_tr)
    c = layout.M2FConfig.from_model_config(cfg)
    specs, total = layout.param_specs(c)
    owners = [s for s in specs if not s.alias_of]
    spans = sorted((s.offset, s.offset + s.numel) for s in owners)
    assert all(a % 64 == 0 for a, _ in spans)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1)), "tensors must not overlap"
    assert spans[-1][1] <= total and total % 64 == 0
    by_name = {s.name: s for s in specs}
    for s in specs:
        if s.alias_of:
            assert by_name[s.alias_of].offset == s.offset and by_name[s.alias_of].numel == s.numel
    assert layout.param_count(c) == sum(int(np.prod(s.shape)) for s in owners)
