"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares, the flat
parameter layout agrees between Python and C, the nn.Module mirror reproduces the reference's state_dict keys
and default initialisation, config validation mirrors the reference, and nothing falls back to CPU compute."""
import ctypes
import os
import re
import sys
import types

import numpy as np
import pytest
import torch
import yaml

import synth
import mer_amd  # noqa: F401
from mer_amd import layout, runtime
from mer_amd.model import M2FNet, FusionAttentionModule

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _shipped_cfg():
    with open(os.path.join(ROOT, "src", "config.yaml")) as f:
        return yaml.safe_load(f)["model"]


def test_config_yaml_keeps_every_reference_key_and_default(golden_dir):
    """src/config.yaml surface: same key names, types and defaults as the reference's file; the only
    additions live under `runtime:`."""
    import json
    with open(os.path.join(golden_dir, "reference_config_keys.json")) as f:
        ref = json.load(f)

    def flatten(d, prefix=""):
        out = {}
        for k, v in d.items():
            out.update(flatten(v, prefix + k + ".") if isinstance(v, dict) else {prefix + k: v})
        return out
    with open(os.path.join(ROOT, "src", "config.yaml")) as f:
        mine = flatten(yaml.safe_load(f))
    for k, v in ref.items():
        assert k in mine, k
        assert type(mine[k]) is type(v) and mine[k] == v, (k, mine[k], v)
    extra = sorted(k for k in mine if k not in ref)
    assert all(k.startswith("runtime.") for k in extra), extra


def test_library_exports_every_declared_symbol():
    header = open(runtime.HEADER_PATH).read()
    declared = set(re.findall(r"\b(m2f_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 20
    l = runtime.lib()
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in include/m2fnet_hip.h but not exported"
    assert declared == set(runtime.SIGNATURES), declared ^ set(runtime.SIGNATURES)


def test_config_struct_size_matches_header():
    assert ctypes.sizeof(runtime.M2FConfigC) == 18 * 4 + 2 * 4


@pytest.mark.parametrize("name", list(synth.CASES) + ["shipped"])
def test_flat_layout_python_equals_c(name):
    cfg = _shipped_cfg() if name == "shipped" else synth.CASES[name][0]
    c = layout.M2FConfig.from_model_config(cfg)
    total = runtime.verify_layout(c)
    specs, t2 = layout.param_specs(c)
    assert total == t2 and all(s.offset % 64 == 0 for s in specs)


def test_param_count_and_flops_match_survey():
    c = layout.M2FConfig.from_model_config(_shipped_cfg())
    assert layout.param_count(c) == 86_251_783
    fwd, fb = layout.flops_per_slot(c, 16)
    assert abs(fwd / 1e6 - 173.07) < 0.01 and abs(fb / 1e6 - 512.15) < 0.01
    c1 = layout.M2FConfig.from_model_config(synth.CASES["c1"][0])
    assert layout.param_count(c1) == 14_382_087
    assert abs(layout.flops_per_slot(c1, 16)[1] / 1e6 - 81.43) < 0.01


def test_workspace_size_query():
    c = layout.M2FConfig.from_model_config(_shipped_cfg())
    cc = runtime.config_to_c(c)
    small = runtime.lib().m2f_workspace_bytes(ctypes.byref(cc), 4, 16, 1)
    big = runtime.lib().m2f_workspace_bytes(ctypes.byref(cc), 32, 16, 1)
    ev = runtime.lib().m2f_workspace_bytes(ctypes.byref(cc), 32, 16, 0)
    assert 0 < small < big and 0 < ev < big
    assert runtime.lib().m2f_workspace_bytes(ctypes.byref(cc), 4, 65, 1) < 0          # L > 64 rejected
    assert b"L" in runtime.lib().m2f_last_error()


def test_state_dict_keys_match_reference_order(golden_dir):
    for name in ("tiny_shared_norm", "tiny_ragged", "tiny_text_only"):
        cfg = synth.CASES[name][0]
        m = M2FNet(cfg)
        keys = list(m.state_dict().keys())
        assert keys == list(synth.make_state_dict(cfg).keys())      # synth order was asserted == reference order
        fx = np.load(os.path.join(golden_dir, name + ".npz"))
        uniq = [k for k, s in zip(keys, layout.param_specs(m.m2f_config)[0]) if not s.alias_of]
        assert uniq == [str(n) for n in fx["grad_names"]]
    m = M2FNet(synth.CASES["tiny_shared_norm"][0])
    assert m.audio_encoders[0].norm is m.audio_encoders[1].norm      # shared, not cloned (model.py:62-65)


def test_default_init_matches_reference_under_same_seed(golden_dir):
    fx = np.load(os.path.join(golden_dir, "init_seed0.npz"))
    cfg = _shipped_cfg()
    cfg = dict(cfg, AUDIO=dict(cfg["AUDIO"], n_encoder_layers=1), TEXT=dict(cfg["TEXT"], n_encoder_layers=2),
               FAM=dict(cfg["FAM"], n_layers=1))
    torch.manual_seed(0)
    m = M2FNet(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(n) for n in fx["names"]]
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    abss = np.array([float(v.double().abs().sum()) for v in sd.values()])
    assert np.allclose(sums, fx["sums"], rtol=0, atol=1e-6) and np.allclose(abss, fx["abs_sums"], rtol=1e-9)


def test_config_validation_mirrors_reference():
    cfg = synth._cfg(64, 64, 64, 4, 4, 4, 1, 1, 1, a_on=False, t_on=False, f_on=False)
    with pytest.raises(ValueError, match="At least one of audio and text must be enabled!"):
        M2FNet(cfg)
    cfg = synth._cfg(64, 64, 64, 4, 4, 4, 1, 1, 1, a_on=False)
    with pytest.raises(ValueError, match="Fusion Attention Module can only be used with both audio and text enabled!"):
        M2FNet(cfg)
    with pytest.raises(AssertionError, match="divisible by num_heads"):
        M2FNet(synth._cfg(300, 64, 64, 8, 4, 4, 1, 1, 1))          # 300 % 8 != 0, like nn.MultiheadAttention


def test_accepts_attribute_style_config():
    def ns(d):
        return types.SimpleNamespace(**{k: (ns(v) if isinstance(v, dict) else v) for k, v in d.items()})
    m = M2FNet(ns(synth.CASES["tiny_ragged"][0]))
    assert isinstance(m.fusion_layers[0], FusionAttentionModule)
    assert m.fusion_layers[0].multihead_attention.in_proj_weight.shape == (192, 64)


def test_no_cpu_fallback():
    cfg, B, L, lengths, kind = synth.CASES["tiny_ragged"]
    text, audio, key_pad, _ = synth.make_inputs(cfg, B, L, lengths, kind)
    m = M2FNet(cfg)
    with pytest.raises(runtime.HipError, match="no CPU fallback"):
        m(text, audio, key_pad)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "multimodal-emotion-recognition_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
    for f in ("model.py", "train.py", "test.py", "dataset.py", "utils.py"):
        src = open(os.path.join(ROOT, "src", f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_dialogue_row_index_matches_dataset_semantics():
    """Dialogues in order of first appearance, utterances sorted by Utterance_ID (reference src/dataset.py:24,35)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "src"))
    from mer_amd.batcher import build_row_index
    from dataset import build_dialogue_index
    dia = [5, 5, 2, 5, 2, 9, 2]
    utt = [1, 0, 2, 2, 0, 0, 1]
    rows = build_row_index(dia, utt)
    assert [r.tolist() for r in rows] == [[1, 0, 3], [4, 6, 2], [5]]
    ids, rows2 = build_dialogue_index(dia, utt)
    assert ids == [5, 2, 9] and [r.tolist() for r in rows2] == [r.tolist() for r in rows]


def test_spill_guard_of_the_gemm_build(tmp_path):
    """csrc/check_spills.py (run by the Makefile on hipcc's resource remarks): a k-contiguous (NT) m2f_gemm16 kernel that
    spills VGPRs must fail the build - its staging loads are issued from inline asm and a spilled register would be reused
    while the load is still in flight; spills in the other forms, and remarks without any such kernel, are reported as such."""
    import subprocess
    script = os.path.join(ROOT, "multimodal-emotion-recognition_amd", "csrc", "check_spills.py")

    def remarks(entries):
        out = []
        for name, spill in entries:
            out.append(f"gemm.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]")
            out.append("gemm.hip:1:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]")
            out.append(f"gemm.hip:1:1: remark:     VGPRs Spill: {spill} [-Rpass-analysis=kernel-resource-usage]")
        p = tmp_path / "r.txt"
        p.write_text("\n".join(out) + "\n")
        return subprocess.run([sys.executable, script, str(p)], capture_output=True, text=True)

    nt = "_ZN12_GLOBAL__N_123m2f_gemm16_dense_kernelILb0ELb0ELi64ELi64ELi128ELi2EEEv9GemmBatch"
    tn = "_ZN12_GLOBAL__N_123m2f_gemm16_dense_kernelILb1ELb1ELi64ELi64ELi128ELi2EEEv9GemmBatch"
    assert remarks([(nt, 0), (tn, 58)]).returncode == 0
    bad = remarks([(nt, 204), (tn, 0)])
    assert bad.returncode == 1 and "spills 204 VGPRs" in bad.stderr
    assert remarks([(tn, 0)]).returncode != 0          # no kernel of the guarded form found: the remark format changed


def test_spill_guard_of_the_ring_builds(tmp_path):
    """check_spills.py --ring (run by the Makefile on every gemm_ring_*.hip): a ring-form kernel waits for its LDS-DMA loads
    with hand-counted vmcnt, and scratch traffic shares that counter - any scratch use must fail the build (a 256x128
    row-major table build with 112 bytes of scratch per lane existed for an hour in round 3)."""
    import subprocess
    script = os.path.join(ROOT, "multimodal-emotion-recognition_amd", "csrc", "check_spills.py")

    def remarks(entries):
        out = []
        for name, scratch in entries:
            out.append(f"./gemm_ring.h:650:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]")
            out.append("./gemm_ring.h:650:1: remark:     VGPRs: 254 [-Rpass-analysis=kernel-resource-usage]")
            out.append(f"./gemm_ring.h:650:1: remark:     ScratchSize [bytes/lane]: {scratch} [-Rpass-analysis=kernel-resource-usage]")
        p = tmp_path / "ring.txt"
        p.write_text("\n".join(out) + "\n")
        return subprocess.run([sys.executable, script, "--ring", str(p)], capture_output=True, text=True)

    a = "_ZN12_GLOBAL__N_122m2f_gemm16_ring_kernelILi256ELi128ELi3ELb1ELb1ELi1EEEv9GemmBatch"
    b = "_ZN12_GLOBAL__N_122m2f_gemm16_ring_kernelILi128ELi128ELi4ELb1ELb1ELi1EEEv9GemmBatch"
    assert remarks([(a, 0), (b, 0)]).returncode == 0
    bad = remarks([(a, 112), (b, 0)])
    assert bad.returncode == 1 and "112 bytes of scratch" in bad.stderr
    assert remarks([("some_other_kernel", 0)]).returncode != 0
    for name in ("gemm_ring_128x128", "gemm_ring_128x64", "gemm_ring_64x64", "gemm_ring_256x128", "gemm_ring_table"):
        built = os.path.join(ROOT, "multimodal-emotion-recognition_amd", "csrc", name + ".remarks")
        if os.path.exists(built):                   # written by the build: the shipped kernels must be clean
            r = subprocess.run([sys.executable, script, "--ring", built], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr


def test_asm_load_checker_flags_a_copy_of_a_register_in_flight():
    """csrc/check_asm_loads.py (run by the Makefile on gemm.hip's ISA): a register that an inline-asm load is still filling
    must not be read, copied or overwritten before the hand-written wait - the pattern that corrupted a build in round 2."""
    import importlib.util
    path = os.path.join(ROOT, "multimodal-emotion-recognition_amd", "csrc", "check_asm_loads.py")
    spec = importlib.util.spec_from_file_location("check_asm_loads", path)
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    good = """kern_good:
	v_sub_u32_e32 v3, v166, v2
	;;#ASMSTART
	global_load_dwordx4 v[70:73], v3, s[18:19]
	;;#ASMEND
	v_sub_u32_e32 v3, v168, v2
	;;#ASMSTART
	global_load_dwordx4 v[74:77], v3, s[18:19] sc1
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(4)
	;;#ASMEND
	ds_write_b128 v9, v[70:73]
	s_endpgm
"""
    bad = good.replace("kern_good", "kern_bad").replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)",
                                                        "\tv_mov_b64_e32 v[30:31], v[74:75]\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)")
    other_block = good.replace("kern_good", "kern_branch").replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)",
                                                                   "\ts_branch .LBB0_9\n.LBB0_3:\n\tv_add_u32_e32 v70, 1, v70\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)")
    note = good.replace("kern_good", "kern_note").replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)",
                                                          "\tv_readfirstlane_b32 s35, v70\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)")
    res = {name: chk.scan(body, name) for name, body in chk.kernels(good + bad + other_block + note)}
    assert set(res) == {"kern_good", "kern_bad", "kern_branch", "kern_note"}
    assert res["kern_good"] == ([], [], True)
    assert len(res["kern_bad"][0]) == 1 and "v_mov_b64" in res["kern_bad"][0][0]
    assert res["kern_branch"][0] == [], "text behind an unconditional branch is another basic block"
    assert res["kern_note"][0] == [] and len(res["kern_note"][1]) == 1
    report = os.path.join(ROOT, "multimodal-emotion-recognition_amd", "csrc", "gemm.asmcheck")
    if os.path.exists(report):                      # written by the build: the shipped kernels must be clean
        assert not [l for l in open(report) if "ERROR" in l]
