#!/usr/bin/env python3
"""Class map of consecutive 128 MiB chunks as igdsp_io_alloc sees it (IGDSP_IO_DEBUG / IGDSP_IO_SURVEY print every probe):
asks for a RECORD + a BULK buffer of <GiB> each so that the walk covers many chunks.  usage: io_survey.py [GiB] [stride]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["IGDSP_IO_DEBUG"] = "1"
os.environ["IGDSP_IO_SURVEY"] = "1"
if len(sys.argv) > 2:
    os.environ["IGDSP_IO_STRIDE"] = sys.argv[2]
import torch
from igate4xsoftphonedsp_amd import capi
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.cuda.set_device(0)
ctx = capi.Context(0, 1024)
B = 128 * 65536 * 160
st, ptrs, rep = ctx.io_alloc([(B, capi.IO_INPUT), (int(gib * 2**30), capi.IO_RECORD), (int(gib * 2**30), capi.IO_BULK)])
print(rep)
st.close(); ctx.close()
