"""bench.py's bookkeeping (no GPU): the per-slot FLOP / byte plan must add up to the network whatever the fusion layout, so that
`roofline.achieved` is priced on the algorithm and not on how many launches it was cut into."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from oracle import yamnet_oracle as O

# profile slots that see a launch per batch (engine.hip run_chunks): slot 2 l + 1 is layer l + 2's pointwise / fused kernel
DEFAULT = [0, 5, 7, 13, 23, 25, 27, 28]                               # layers 5-7 one launch (slot 13), layers 8-12 + depthwise 13 one (slot 23)
NO_MID = [0, 5, 7, 9, 11, 13, 23, 25, 27, 28]                         # bd_set_fusion separable = 10: layers 5-7 on their four kernels
RUN_TO_11 = [0, 5, 7, 9, 11, 13, 21, 23, 25, 27, 28]                  # bd_set_fusion separable = 7: the run ends at layer 11
PER_LAYER = [0, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 28]      # a launch per layer (the bookkeeping must still add up)


def _plan(slots, chip=True, tail=False):
    launches = np.zeros(29, dtype=np.int64)
    launches[slots] = 40
    return bench.slot_plan(launches, chip=chip, tail=tail)


def _network_flops():
    """2 x MACs of conv1, every depthwise and every pointwise of a 96 x 64 patch (yamnet.py:26-106) + pool + head."""
    h, w, c = 48, 32, 32
    total = 2 * 9 * h * w * c
    for stride, cout in O.LAYER_DEFS[1:]:
        h, w = h // stride, w // stride
        total += 2 * 9 * h * w * c + 2 * h * w * c * cout
        c = cout
    return total + h * w * c + 2 * 1024 * 13


def test_slot_plan_adds_up_to_the_network_in_both_layouts():
    for slots in (DEFAULT, NO_MID, RUN_TO_11, PER_LAYER):
        for chip in (True, False):
            if slots is DEFAULT and not chip:
                continue
            plan = _plan(slots, chip)
            assert sorted(plan) == sorted(slots)
            assert sum(v[3] for v in plan.values()) == _network_flops()
    a, b = _plan(RUN_TO_11, chip=False), _plan(PER_LAYER)
    assert sum(v[2] for v in a.values()) == sum(v[2] for v in b.values())       # round-3 form of the run: it still stores every layer
    # the on-chip run (sepchip.hip): the same FLOP, and of the four layers' 8 x 49 152 B per window only the run's input and
    # output are algorithmic traffic
    c = _plan(RUN_TO_11)
    assert c[21][:2] == ("sep8-11", "sep_chip_kernel") and c[21][2] == 2 * 24 * 512 * 4
    assert sum(v[2] for v in a.values()) - sum(v[2] for v in c.values()) == 6 * 24 * 512 * 4
    # ... and with layer 12 + depthwise 13 along (the default): [24][512] in, [6][512] out per window
    d = _plan(NO_MID)
    assert d[23][:2] == ("sep8-12+dw13", "sep_chip_kernel") and d[23][2] == (24 + 6) * 512 * 4
    assert d[23][3] == c[21][3] + c[23][3]
    # ... and pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as one launch: [96][128] in, [24][512] out per window
    m = _plan(DEFAULT)
    assert m[13][:2] == ("pw5-pw7", "sep_mid_kernel") and m[13][2] == (96 * 128 + 24 * 512) * 4
    assert m[13][3] == d[9][3] + d[11][3] + d[13][3] and m[7] == d[7]
    # ... and layers 13 / 14 on septail.hip's kernel (the default): depthwise 14 moves from layer 14's slot into pointwise 13's;
    # [6][512] in and [6][1024] out per window (as f16 planes: 4 bytes per element), then [6][1024] in and [1024] out
    t = _plan(DEFAULT, tail=True)
    assert sorted(t) == sorted(DEFAULT) and sum(v[3] for v in t.values()) == _network_flops()
    assert t[25][:2] == ("pw13+dw14", "tail_gemm_kernel") and t[25][2] == (6 * 512 + 6 * 1024) * 4
    assert t[27][:2] == ("pw14+pool", "tail_gemm_kernel") and t[27][2] == (6 * 1024 + 1024) * 4 and t[28][0] == "head"
    assert t[25][3] + t[27][3] == m[25][3] + m[27][3] and t[23] == m[23]


def test_the_run_and_the_next_depthwise_forms_are_families_of_their_own():
    plan = _plan(RUN_TO_11, chip=False)
    assert plan[21][:2] == ("sep8-11", "sep_w12_kernel")
    assert plan[21][3] == 4 * _plan(PER_LAYER)[15][3]
    assert plan[23][:2] == ("sep12+dw13", "sep_w12_ndw_kernel")                 # 512 -> 512 with layer 13's depthwise: 12-wave kernel
    assert plan[11][:2] == ("sep6+dw7", "sep_ws_kernel")                        # 256 -> 256: one column tile, 8-wave kernel
    assert plan[27][:2] == ("sep14+pool", "sep_w12_ndw_kernel") and plan[28][0] == "head"     # two 512-column halves + pool
    assert plan[25][:2] == ("pw13", "sep_ws_kernel")


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
