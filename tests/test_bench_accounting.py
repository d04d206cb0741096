"""bench.py's bookkeeping (no GPU): the per-slot FLOP / byte plan must add up to the network whatever the fusion layout, so that
`roofline.achieved` is priced on the algorithm and not on how many launches it was cut into."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from oracle import yamnet_oracle as O

# profile slots that see a launch per batch (engine.hip run_chunks): slot 2 l + 1 is layer l + 2's pointwise / fused kernel
DEFAULT = [0, 5, 7, 13, 23, 25, 27, 28]                               # layers 5-7 one launch (slot 13), layers 8-12 + depthwise 13 one (slot 23), the tail (25, 27)
NO_MID = [0, 5, 7, 9, 11, 13, 23, 25, 27, 28]                         # bd_set_fusion separable = 10: layers 5-7 on their four kernels
PER_OP = list(range(0, 29))                                           # bd_set_fusion 0 / 0: conv1, every depthwise, every 1x1, pool + head


def _plan(slots):
    launches = np.zeros(29, dtype=np.int64)
    launches[slots] = 40
    return bench.slot_plan(launches)


def _network_flops():
    """2 x MACs of conv1, every depthwise and every pointwise of a 96 x 64 patch (yamnet.py:26-106) + pool + head."""
    h, w, c = 48, 32, 32
    total = 2 * 9 * h * w * c
    for stride, cout in O.LAYER_DEFS[1:]:
        h, w = h // stride, w // stride
        total += 2 * 9 * h * w * c + 2 * h * w * c * cout
        c = cout
    return total + h * w * c + 2 * 1024 * 13


def test_slot_plan_adds_up_to_the_network_in_every_launch_set():
    for slots in (DEFAULT, NO_MID, PER_OP):
        plan = _plan(slots)
        assert sorted(plan) == sorted(slots)
        assert sum(v[3] for v in plan.values()) == _network_flops()
    # the on-chip run (sepchip.hip) with layer 12 + depthwise 13 along: of the five layers' tiles only the run's input and output are
    # algorithmic traffic - [24][512] in, [6][512] out per window
    d = _plan(NO_MID)
    assert d[23][:2] == ("sep8-12+dw13", "sep_chip_kernel") and d[23][2] == (24 + 6) * 512 * 4
    o = _plan(PER_OP)
    assert d[23][3] == sum(o[s_][3] for s_ in range(14, 25))                      # depthwise 8 .. depthwise 13
    # ... pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as one launch: [96][128] in, [24][512] out per window
    m = _plan(DEFAULT)
    assert m[13][:2] == ("pw5-pw7", "sep_mid_kernel") and m[13][2] == (96 * 128 + 24 * 512) * 4
    assert m[13][3] == d[9][3] + d[11][3] + d[13][3] and m[7] == d[7]
    # ... layers 13 / 14 on septail.hip's kernel: depthwise 14 is pointwise 13's epilogue; [6][512] in and [6][1024] out per window
    # (as f16 planes: 4 bytes per element), then [6][1024] in and [1024] out
    assert m[25][:2] == ("pw13+dw14", "tail_gemm_kernel") and m[25][2] == (6 * 512 + 6 * 1024) * 4
    assert m[27][:2] == ("pw14+pool", "tail_gemm_kernel") and m[27][2] == (6 * 1024 + 1024) * 4 and m[28][0] == "head"
    assert m[25][3] + m[27][3] == o[25][3] + o[26][3] + o[27][3] + 6 * 1024 and m[23] == d[23]


def test_the_kernels_of_the_other_launch_sets_are_named():
    d = _plan(NO_MID)
    assert d[7][:2] == ("sep4+dw5", "l4_window_kernel") and d[9][:2] == ("pw5", "pw_res_kernel")
    assert d[11][:2] == ("sep6+dw7", "sep_ws_kernel") and d[13][:2] == ("pw7", "pw_res_kernel")
    o = _plan(PER_OP)
    assert o[1][1] == "conv1_kernel" and o[2][1] == "depthwise_kernel" and o[28][0] == "pool_head"
    assert o[3][1] == "pointwise_f16x3_kernel" and o[25][1] == "sep_ws_kernel"      # 32 -> 64 on the tile kernel, 512 -> 1024 wave-specialised


def test_strict_f32_is_measured_like_the_headline():
    """value_strict_f32 comes from the SAME run_files() loop as `value` (asserted on the source: the line
    needs a GPU), says so with the same workload description, and carries a kernel-event roofline block of its own; the
    exact-f32 mode's slot plan adds up to the network like the default one."""
    src = open(bench.__file__).read()
    assert 's_elapsed = timed_region(strict_steps)' in src and 'e.set_pointwise_mode("f32")' in src
    assert '"workload": out["config"]["workload"]' in src                       # one workload description for both values
    for key in ('"roofline_strict"', '"strict_f32"', '"avg_launch_us"', '"flop_per_launch_avg"', 'PEAK_F32_MFMA_TFLOPS', '"traffic"',
                'PMC_TRAFFIC_STRICT_FILE'):
        assert key in src, key
    launches = np.zeros(29, dtype=np.int64)
    f32_slots = [0, 5, 7] + list(range(9, 28, 2)) + [28]
    launches[f32_slots] = 40
    plan = bench.slot_plan_f32(launches)
    assert sorted(plan) == f32_slots
    assert sum(v[3] for v in plan.values()) == _network_flops()
    assert plan[5][1] == "stem_reg_f32_kernel" and plan[7][1] == "l4_reg_f32_kernel" and plan[27][0] == "pw14+pool"
    assert all(plan[s][1] == "pointwise_kernel" for s in range(9, 28, 2))
    # the default launch set since round 5: layers 8-12 + depthwise 13 as ONE launch in layer 12's slot (sepchipf32.hip)
    launches[:] = 0
    chip_slots = [0, 5, 7, 9, 11, 13, 23, 25, 27, 28]
    launches[chip_slots] = 40
    plan = bench.slot_plan_f32(launches)
    assert sorted(plan) == chip_slots
    assert sum(v[3] for v in plan.values()) == _network_flops()
    assert plan[23][:2] == ("sep8-12+dw13", "sep_chip_f32_kernel") and plan[13][0] == "pw7+dw8" and plan[25][0] == "pw13+dw14"
    # ... and pointwise 5 + layers 6-7 as ONE launch in layer 7's slot (sepmidf32.hip): depthwise 8 moves into the run behind it
    launches[:] = 0
    mid_slots = [0, 5, 7, 13, 23, 25, 27, 28]
    launches[mid_slots] = 40
    plan = bench.slot_plan_f32(launches)
    assert sorted(plan) == mid_slots
    assert sum(v[3] for v in plan.values()) == _network_flops()
    assert plan[13][:2] == ("pw5-pw7", "sep_mid_f32_kernel") and plan[23][1] == "sep_chip_f32_kernel"


def test_the_line_says_what_was_measured():
    """the keys of the JSON line that name the arithmetic and the scope of `value` (asserted on
    the source: the line itself needs a GPU; the driver's BENCH record carries it)."""
    src = open(bench.__file__).read()
    for key in ('"dtype": "f32 (split-f16x3 MFMA products, f32 accumulate)"', '"value_strict_f32"', '"value_end_to_end_f32_host"',
                '"pipeline_frac"', '"timed_read_back"', '"ranks_seen"', '"valu_floor_frac"', '"resample_roofline"',
                '"resample_roofline_scipy"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key
    assert bench.PEAK_F16_MFMA_TFLOPS == 2500.0 and abs(bench.PEAK_SPLIT_F16_TFLOPS - 2500.0 / 3) < 1e-9
    # pipeline_frac at round 3's driver-observed value: 1.609 M windows/s x 132.12 MFLOP x 3 = 638 TFLOP/s = 0.255
    assert abs(3.0 * bench.POINTWISE_FLOP_PER_WINDOW * 1.609e6 / 1e12 / bench.PEAK_F16_MFMA_TFLOPS - 0.255) < 1e-3
