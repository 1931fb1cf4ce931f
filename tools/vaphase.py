"""Cycles a wave of k_view_attn spends per region of its tile / view loops (a -DGPN_STAMP build):
python tools/vaphase.py <lib.so> [fp32|bf16x3].  Regions (per wave of workgroup 3, HOMOG view loop only):
tile prologue | wait + ring read-back | DMA refill | MFMA section | side store + softmax | tile epilogue."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPN_LIB"] = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:3] + ["one"]
import runpy

try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vabench.py"), run_name="__main__")
except SystemExit:
    pass
h = C.CDLL(os.environ["GPN_LIB"])
buf = (C.c_ulonglong * 128)()
h.gpn_debug_ra_phase(buf)
v = list(buf)
names = ["prologue", "wait+read", "dma", "mfma", "softmax", "epilogue"]
for w in range(8):
    row = v[w * 8: w * 8 + 8]
    it, tiles = max(1, row[6]), max(1, row[7])
    print("wave %d: prologue %6d /tile  wait+read %5d  dma %5d  mfma %5d  softmax %5d /view   epilogue(+other loops) %6d /tile   (%d views, %d tiles, total %d cycles)"
          % (w, row[0] // tiles, row[1] // it, row[2] // it, row[3] // it, row[4] // it, row[5] // tiles, row[6], row[7], sum(row[:6])))
