"""SURVEY 8(f) rank 4 — G.726 code-word reorder (changeUplinkOrder, roip_ed137.cpp:6379-6499).
CPU: properties of the oracle's bug-for-bug restatement.  GPU: kernel == oracle, bit-exact."""
import numpy as np
import pytest

from igate4xsoftphonedsp_amd import capi


def test_oracle_properties(orc):
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, 3000, dtype=np.uint8)
    # modes 1 and 3 are involutions (field reversal / nibble swap)
    assert np.array_equal(orc.g726_reorder(orc.g726_reorder(d, 1), 1), d)
    assert np.array_equal(orc.g726_reorder(orc.g726_reorder(d, 3), 3), d)
    assert orc.g726_reorder(np.array([0b11100100], np.uint8), 1)[0] == 0b00011011
    assert orc.g726_reorder(np.array([0xA5], np.uint8), 3)[0] == 0x5A
    # mode 2 is a bijection on 24-bit groups: all sample codes survive, just in another order
    g = np.zeros(3, np.uint8)
    seen = set()
    for v in rng.integers(0, 1 << 24, 500):
        g[:] = [v & 255, (v >> 8) & 255, (v >> 16) & 255]
        o = orc.g726_reorder(g, 2)
        seen.add(int(o[0]) | int(o[1]) << 8 | int(o[2]) << 16)
    assert len(seen) == 500
    # S1 (the first 3-bit code, V bits 0-2) lands in the top three bits of output byte 0
    assert orc.g726_reorder(np.array([0b101, 0, 0], np.uint8), 2).tolist() == [0b101 << 5, 0, 0]
    # mode 4: the reference's S2_ field is always zero -> output byte 1 bits 6-7 never set (documented quirk)
    o = orc.g726_reorder(rng.integers(0, 256, 5 * 400, dtype=np.uint8), 4).reshape(-1, 5)
    assert np.all((o[:, 1] & 0xC0) == 0)
    assert orc.g726_reorder(np.array([0x1F, 0, 0, 0, 0], np.uint8), 4).tolist() == [0x1F << 3, 0, 0, 0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_gpu_matches_oracle(orc, mode):
    from tests import gpu_util as gu

    torch = gu.torch_cuda()
    ctx = capi.Context(0, 16)
    group = {1: 1, 2: 3, 3: 1, 4: 5}[mode]
    for n_groups in (1, 7, 64, 4099, 100000):
        n = n_groups * group
        data = orc.gen_uniform(n, seed=mode * 1000 + n_groups)
        d_out = gu.dev_zeros(n + 32, 0xEE)
        ctx.g726_reorder(gu.to_dev(data), d_out, n, mode)
        torch.cuda.synchronize()
        got = gu.to_host(d_out, np.uint8)
        assert np.array_equal(got[:n], orc.g726_reorder(data, mode)), (mode, n)
        assert np.all(got[n:] == 0xEE)
    # typical 20 ms G.726 payloads: 40 / 60 / 80 / 100 bytes per frame, 4096 frames
    n = {1: 40, 2: 60, 3: 80, 4: 100}[mode] * 4096
    data = orc.gen_uniform(n, seed=99)
    d_out = gu.dev_zeros(n)
    ctx.g726_reorder(gu.to_dev(data), d_out, n, mode)
    torch.cuda.synchronize()
    assert np.array_equal(gu.to_host(d_out, np.uint8), orc.g726_reorder(data, mode))
    if group > 1:
        assert ctx.L.igdsp_g726_reorder(ctx.h, d_out.data_ptr(), d_out.data_ptr(), group + 1, mode, None) == -22
    assert ctx.L.igdsp_g726_reorder(ctx.h, d_out.data_ptr(), d_out.data_ptr(), 15, 9, None) == -22
    ctx.close()
