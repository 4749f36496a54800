"""The eight-phase 256x256 bf16 GEMM (csrc/gemm_p8.h: the weight-gradient table launch's kernel since round 4, and the text encoder's
large forward launches) through its kernel-level C entry point m2f_gemm_p8 - the checks of tools/p8_bench.py:
  * both operand forms (row-major RC = dW = dY^T X with bias-gradient row sums and ReLU on the X operand; k-contiguous KC = nn.Linear
    forward with bias / ReLU / GELU / residual) against fp64 products of the SAME bf16 operands, on whole tiles, ragged edges (300- and
    7-wide operands, 252 columns), k tails (48, 1000 rows of reduction), padded leading dimensions, results with a row stride;
  * the same launch repeated, with the tiles dealt to 256 and to 7 workgroups (several tiles per workgroup = the continuous prefetch
    stream across tile boundaries, the relaxed first wait behind an epilogue): every result bit-identical to the first.
A hand-placed LDS-DMA / barrier schedule passes or fails by its counted waits, not by luck: the repeated runs are the race screen the
guide asks for after a change to a synchronisation structure."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _tool():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import p8_bench
    return p8_bench


def test_p8_matches_fp64_products_of_the_same_bf16_operands():
    _tool().check()


def test_p8_results_do_not_depend_on_the_run_or_on_the_tile_distribution():
    _tool().race(12)


def test_relu_on_the_a_operand_is_refused_not_ignored():
    import torch
    pb = _tool()
    a, b = pb.operands(1, 256, 256, 64, 1)
    c = torch.empty(256, 256, device="cuda")
    with pytest.raises(RuntimeError):
        pb.run(1, 256, 256, 64, a.view(torch.int16), b.view(torch.int16), c, relu_a=1)
