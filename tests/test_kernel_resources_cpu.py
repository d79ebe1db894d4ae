"""-m "not gpu": no hot kernel may spill.  Builds the gfx950 code objects with `--asm` (hipcc cross-compiles without a GPU) and
reads the kernel descriptors out of the saved ISA: `.vgpr_spill_count` and `.private_segment_fixed_size` must be 0 for every
kernel that touches the payload stream (round 1 shipped the headline instantiation with 3 spilled VGPRs = 16 B of scratch per
lane), and the LDS / VGPR budgets the launch geometry relies on must hold."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = ("k_meter_chunk64", "k_meter_rtp64", "k_meter_image", "k_meter_strided", "k_meter_tiny", "k_meter_wave_per_frame", "k_roundtrip_lut64", "k_roundtrip_blk64", "k_roundtrip_strided", "k_roundtrip_chunk64",
       "k_roundtrip_general", "k_encode_lut16", "k_encode_v8", "k_depayload64", "k_wav_expand16", "k_flush_fold", "k_window_update", "k_window_finish")


@pytest.fixture(scope="module")
def resources():
    from igate4xsoftphonedsp_amd import build as b
    import kernel_resources as kr

    srcs = [os.path.join(b.CSRC, s) for s in b.DEVICE_SOURCES] + [os.path.join(b.CSRC, h) for h in ("igdsp_internal.h", "igdsp_device.h")]
    if len(kr.asm_files()) < 4 or any(os.path.getmtime(s) > min(os.path.getmtime(a) for a in kr.asm_files()) for s in srcs):
        b.build(save_asm=True)
    return kr.resources()


def test_no_hot_kernel_spills(resources):
    seen = set()
    for r in resources:
        name = r["demangled"]
        if not any(h in name for h in HOT) or "DIAG" in name:
            continue
        if "k_meter_chunk64<false, false, true>" in name:          # the cycle-stamp diagnostic instantiation
            continue
        seen.add(next(h for h in HOT if h in name))
        assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
    assert seen == set(HOT), set(HOT) - seen


def test_launch_geometry_budgets(resources):
    by = {r["demangled"]: r for r in resources}
    lim = {"k_meter_chunk64<false, true, false>": 128, "k_meter_chunk64<false, false, false>": 128,      # 16 waves / CU
           "k_meter_chunk64<true, true, false>": 168, "k_meter_chunk64<true, false, false>": 168,        # 12 waves / CU
           "k_roundtrip_lut64<0>": 168, "k_roundtrip_lut64<1>": 168, "k_encode_lut16<0>": 128, "k_encode_lut16<1>": 128,
           "k_roundtrip_blk64<0>": 168, "k_roundtrip_blk64<1>": 168,                                      # up to 12 waves / CU (6 launched)
           "k_meter_rtp64<true, false, false, 2>": 170, "k_meter_rtp64<true, false, true, 2>": 170, "k_meter_rtp64<true, true, false, 2>": 170}   # 12 waves / CU
    for k, v in lim.items():
        r = by["void igdsp::" + k]
        assert r["vgpr"] <= v, (k, r)
        assert r["lds"] <= 160 * 1024, (k, r)
