"""CPU-side checks of the drop-in boundary: libigdsp.so loads without a GPU, exports every
symbol include/igdsp.h declares, the struct layouts agree between header, binding and oracle,
and the product fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from igate4xsoftphonedsp_amd import build as igbuild
from igate4xsoftphonedsp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    igbuild.build()
    return capi.load()


def test_header_symbols_all_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "igdsp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(igdsp_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in capi.PROTOTYPES}
    assert declared == bound, (declared ^ bound)
    raw = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.igdsp_abi_version() == capi.ABI_VERSION == 3


def test_struct_layouts_agree(orc):
    assert capi.FRAME_STATS.itemsize == orc.FRAME_STATS.itemsize == 16
    assert capi.CHAN_HOLD.itemsize == orc.CHAN_HOLD.itemsize == 32
    assert capi.AGGREGATE.itemsize == orc.AGGREGATE.itemsize == 8 * capi.AGG_WORDS == 896
    assert ctypes.sizeof(capi.Level) == 20
    # ABI 3: the gated window
    assert capi.CHAN_PROBE.itemsize == orc.CHAN_PROBE.itemsize == ctypes.sizeof(capi.ChanProbe) == 8
    assert ctypes.sizeof(capi.Window) == 8 + 4 * ctypes.sizeof(ctypes.c_void_p) == 40
    assert (capi.GATE_ALWAYS, capi.GATE_SQU, capi.GATE_PTT, capi.GATE_SQU_OR_PTT) == (orc.GATE_ALWAYS, orc.GATE_SQU, orc.GATE_PTT, orc.GATE_SQU_OR_PTT) == (0, 1, 2, 3)
    for a, b in ((capi.FRAME_STATS, orc.FRAME_STATS), (capi.CHAN_HOLD, orc.CHAN_HOLD), (capi.AGGREGATE, orc.AGGREGATE)):
        assert a.names == b.names
        assert [a.fields[n][1] for n in a.names] == [b.fields[n][1] for n in b.names]
    # header constants mirrored in the binding
    hdr = open(os.path.join(ROOT, "include", "igdsp.h")).read()
    for name, val in (("IGDSP_PT_PCMU", capi.PT_PCMU), ("IGDSP_PT_PCMA", capi.PT_PCMA), ("IGDSP_PT_R2S", capi.PT_R2S),
                      ("IGDSP_SAMPLES_PER_FRAME", capi.SAMPLES_PER_FRAME), ("IGDSP_MAX_PAYLOAD", capi.MAX_PAYLOAD),
                      ("IGDSP_AGG_MAX_RANKS", capi.AGG_MAX_RANKS), ("IGDSP_PROBE_ALARM", capi.PROBE_ALARM), ("IGDSP_PKT_MIXED", capi.PKT_MIXED),
                      ("IGDSP_GATE_SQU_OR_PTT", capi.GATE_SQU_OR_PTT), ("IGDSP_ABI_VERSION", capi.ABI_VERSION)):
        m = re.search(rf"#define\s+{name}\s+(\d+)", hdr)
        assert m and int(m.group(1)) == val, name


def test_no_cpu_fallback_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.igdsp_create(ctypes.byref(h), 0, 16) == -19        # IGDSP_ENODEV
    assert not h.value
    with pytest.raises(capi.IgdspError):
        capi.Context(device=0, max_channels=16)
    # NULL-context calls are rejected, never computed on the host
    assert lib.igdsp_decode_meter(None, None, None, None, 1, 1, 160, None, None, None, 0, None) == -22
    assert lib.igdsp_flush(None, None) == -22
    assert lib.igdsp_destroy(None) == 0


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "igate4xsoftphonedsp_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_asm" in dirpath or "__pycache__" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "igdsp_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
    so = open(capi.LIB_PATH, "rb").read()
    assert b"orc_" not in so
