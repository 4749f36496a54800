"""Per-launch floor of the chain GEMMs (forward + input-gradient launches) of one bf16 step against what they take.

For every grouped GEMM launch of the plan (m2f_plan_gemm_shapes) the tile configuration the launcher picks (gemm.hip::launch_tile16:
256x128 for two-round launches, 128x128 from 200 tiles, 128x64 from 150, 64x64 from 80, register-staged 64x64 below) and two bounds
per workgroup tile, in cycles:
    mfma   = BM BN K / 2048        (4 SIMDs x 1,024 bf16 FLOP per clock, both waves of a SIMD sharing its pipe)
    ingest = (BM + BN) K 2 bytes / R,  R = the LDS-DMA rate of a CU: 40 B/clk measured with <= 128 CUs streaming, 30 with all 256
             
    floor  = rounds x max(mfma, ingest) / clock + 1.5 us of dependent-kernel boundary (MI355X_MICROARCH.md price list, `boundary`)
summed over the step, beside the measured hipEvent time of the same launch (m2f_step_timed; an event pair adds ~2.2 us of its own).
usage: python tools/chain_floor.py [workload]   (default c3)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import bench  # noqa: E402
import mer_amd  # noqa: E402,F401
from mer_amd import runtime  # noqa: E402
from mer_amd.model import M2FNet  # noqa: E402

CLOCK_GHZ = 2.1          # what chain launches hold (they are far from MFMA-dense; the dense table launch runs at ~1.6-1.7)


def pick_tile(probs):
    cnt = lambda bm, bn: sum(-(-M // bm) * -(-N // bn) for M, N, K in probs)
    t128 = cnt(128, 128)
    if 256 < t128 < 512 and cnt(256, 128) <= 256:
        return 256, 128, "ring"
    if t128 >= 200:
        return 128, 128, "ring"
    if cnt(128, 64) >= 150:
        return 128, 64, "ring"
    if cnt(64, 64) >= 80:
        return 64, 64, "ring"
    return 64, 64, "register-staged"


def main():
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
    cfg, B, L = wl["cfg"], wl["B"], wl["L"]
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="bf16", shape_buckets=False).cuda().train()
    text, audio, key_pad, emotion = bench.synthetic_batch(cfg, B, L, 0, torch.device("cuda"), False)
    plan = m.engine().plan(B, L, True, True)
    plan.set_inputs(text, audio, key_pad, emotion)
    fn = runtime.lib().m2f_plan_gemm_shapes
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    buf = (ctypes.c_int * 8192)()
    n = fn(plan.handle, buf, 8192)
    assert n > 0, n
    launches, i = [], 0
    while i < n:
        phase, layout, group, count = buf[i: i + 4]
        probs = [tuple(buf[i + 4 + 3 * k: i + 7 + 3 * k]) for k in range(count)]
        launches.append((phase, layout, group, probs))
        i += 4 + 3 * count
    for _ in range(5):
        plan.step_timed(0.1, False, False)
    reps = [plan.step_timed(0.1, False, False) for _ in range(10)]
    ms = [sum(r[j][1] for r in reps) / len(reps) for j in range(len(reps[0]))]
    kinds = [reps[0][j][0] & 31 for j in range(len(reps[0]))]
    gemm_ms = [ms[j] for j in range(len(ms)) if kinds[j] in (0, 1)]          # forward / input-gradient forms, in launch order
    assert len(gemm_ms) == len(launches), (len(gemm_ms), len(launches))
    print(f"# {wl['name']}: chain GEMM launches of one bf16 step (forward + input gradient), floor model in the docstring of tools/chain_floor.py")
    print(f"# {'#':>3s} {'part':10s} {'problems (M x N x K)':44s} {'tile':>8s} {'tiles':>5s} {'rnd':>3s} {'mfma':>6s} {'ingest':>6s} {'floor us':>8s} {'event us':>8s} {'x floor':>7s}")
    tot_floor = tot_meas = tot_mfma = 0.0
    parts = ["encoders", "fusion", "classifier"]
    by_part = {}
    for j, ((phase, layout, group, probs), t_ms) in enumerate(zip(launches, gemm_ms)):
        bm, bn, form = pick_tile(probs)
        tiles = sum(-(-M // bm) * -(-N // bn) for M, N, K in probs)
        rounds = -(-tiles // 256)
        kmax = max(K for _, _, K in probs)
        rate = 40.0 if tiles <= 128 else 30.0
        mfma = bm * bn * kmax / 2048.0
        ingest = (bm + bn) * kmax * 2 / rate
        floor_us = rounds * max(mfma, ingest) / (CLOCK_GHZ * 1e3) + 1.5
        flop = sum(2.0 * M * N * K for M, N, K in probs)
        mfma_only_us = flop / 2.5e15 * 1e6
        meas = t_ms * 1e3
        tot_floor += floor_us; tot_meas += meas; tot_mfma += mfma_only_us
        d = by_part.setdefault(parts[group], [0.0, 0.0, 0])
        d[0] += floor_us; d[1] += meas; d[2] += 1
        desc = " + ".join(f"{M}x{N}x{K}" for M, N, K in probs)
        print(f"  {j:3d} {('fwd ' if phase == 0 else 'bwd ') + parts[group][:5]:10s} {desc[:44]:44s} {f'{bm}x{bn}':>8s} {tiles:5d} {rounds:3d} {mfma:6.0f} {ingest:6.0f} {floor_us:8.2f} {meas:8.2f} {meas / floor_us:7.2f}")
    print(f"# sum over {len(launches)} launches: floor {tot_floor / 1e3:.3f} ms, measured (hipEvent intervals, ~2.2 us each above the kernels' own time) {tot_meas / 1e3:.3f} ms "
          f"= {tot_meas / tot_floor:.2f} x the floor; at the 2.5 PFLOP/s peak alone the same FLOPs take {tot_mfma / 1e3:.3f} ms")
    for k, (f, me, c) in by_part.items():
        print(f"#   {k:10s}: {c:3d} launches, floor {f:7.1f} us, measured {me:7.1f} us ({me / f:.2f} x)")


if __name__ == "__main__":
    runtime.require_gpu()
    main()
