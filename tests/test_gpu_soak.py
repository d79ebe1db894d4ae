"""-m gpu: a short soak — 150 back-to-back calls mixing every batched entry point, shapes and streams on ONE
context, each checked against the oracle.  Guards against state leaking between launches (LDS tables, work
queues, aggregate slots, hold windows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from igate4xsoftphonedsp_amd import capi  # noqa: E402
from tests import gpu_util as gu  # noqa: E402


def test_mixed_entry_points_back_to_back(orc):
    torch = gu.torch_cuda()
    ctx = capi.Context(0, 256)
    rng = np.random.default_rng(77)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    hold_gpu, hold_ref = None, None
    C_h = 128
    for it in range(150):
        kind = int(rng.integers(0, 5))
        s = streams[it & 1]
        with torch.cuda.stream(s):
            hs = s.cuda_stream
            if kind == 0:                                              # tuned meter path (+ tail), with aggregate
                C_, F_ = int(rng.choice([64, 96, 200, 1024])), int(rng.integers(1, 6))
                pl = orc.gen_uniform(F_ * C_ * 160, seed=it).reshape(F_, C_, 160)
                cd = rng.choice(np.array([0, 8], np.uint8), size=C_)
                d_st, d_agg = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(capi.AGGREGATE.itemsize)
                ctx.decode_meter(gu.to_dev(pl), gu.to_dev(cd), C_, F_, 160, d_st, agg=d_agg, rank=it % 8, stream=hs)
                s.synchronize()
                est, eagg = orc.decode_meter(pl, cd, want_agg=True, rank=it % 8)
                gu.assert_stats_equal(gu.to_host(d_st, capi.FRAME_STATS, (F_, C_)), est, n=160)
                assert gu.to_host(d_agg, capi.AGGREGATE)[0].tobytes() == eagg.tobytes()
            elif kind == 1:                                            # encode, both lineages
                C_, F_, n = 7, 3, int(rng.choice([160, 24, 33]))
                pcm = rng.integers(-32768, 32768, size=(F_, C_, n)).astype("<i2")
                cd = rng.choice(np.array([0, 8], np.uint8), size=C_)
                v = it & 1
                d_out = gu.dev_zeros(F_ * C_ * n, 0xEE)
                ctx.encode(gu.to_dev(pcm), gu.to_dev(cd), C_, F_, n, d_out, variant=v, stream=hs)
                s.synchronize()
                assert np.array_equal(gu.to_host(d_out, np.uint8, (F_, C_, n)), orc.encode(pcm, cd, v))
            elif kind == 2:                                            # fused round trip with a persistent hold window
                F_ = int(rng.choice([2, 8, 24]))
                cd = np.where(np.arange(C_h) & 1, 8, 0).astype(np.uint8)
                pl = orc.gen_uniform(F_ * C_h * 160, seed=1000 + it).reshape(F_, C_h, 160)
                if hold_gpu is None:
                    hold_gpu, hold_ref = gu.to_dev(gu.new_hold(C_h)), orc.hold_new(C_h)
                d_out, d_st = gu.dev_zeros(F_ * C_h * 160), gu.dev_zeros(F_ * C_h * 16)
                ctx.roundtrip_peakhold(gu.to_dev(pl), gu.to_dev(cd), C_h, F_, 160, d_out, d_st, hold_gpu, variant=it & 1, stream=hs)
                s.synchronize()
                eout, _, hold_ref = orc.roundtrip_peakhold(pl, cd, hold_ref, variant=it & 1)
                assert np.array_equal(gu.to_host(d_out, np.uint8, (F_, C_h, 160)), eout)
                assert gu.to_host(hold_gpu, capi.CHAN_HOLD).tobytes() == hold_ref.tobytes()
            elif kind == 3:                                            # general path: ragged, odd n, PCM out
                C_, F_, n = int(rng.integers(1, 40)), int(rng.integers(1, 5)), int(rng.choice([1, 24, 159, 164, 256]))
                pl = orc.gen_uniform(F_ * C_ * n, seed=2000 + it).reshape(F_, C_, n)
                cd = rng.choice(np.array([0, 8], np.uint8), size=C_)
                ln = rng.integers(0, n + 1, size=(F_, C_)).astype(np.uint16)
                d_st, d_pcm = gu.dev_zeros(F_ * C_ * 16, 0xEE), gu.dev_zeros(F_ * C_ * n * 2, 0xEE)
                ctx.decode_meter(gu.to_dev(pl), gu.to_dev(cd), C_, F_, n, d_st, pcm=d_pcm, length=gu.to_dev(ln), stream=hs)
                s.synchronize()
                est, epcm = orc.decode_meter(pl, cd, length=ln, want_pcm=True)
                gu.assert_stats_equal(gu.to_host(d_st, capi.FRAME_STATS, (F_, C_)), est, n=ln)
                assert np.array_equal(gu.to_host(d_pcm, "<i2", (F_, C_, n)), epcm)
            else:                                                      # G.726 reorder
                mode = int(rng.integers(1, 5))
                g = {1: 1, 2: 3, 3: 1, 4: 5}[mode]
                nb = g * int(rng.integers(1, 5000))
                data = orc.gen_uniform(nb, seed=3000 + it)
                d_out = gu.dev_zeros(nb)
                ctx.g726_reorder(gu.to_dev(data), d_out, nb, mode, stream=hs)
                s.synchronize()
                assert np.array_equal(gu.to_host(d_out, np.uint8), orc.g726_reorder(data, mode))
    assert ctx.L.igdsp_last_error(ctx.h) in (b"", None)
    ctx.close()
