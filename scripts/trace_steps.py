#!/usr/bin/env python3
"""Several consecutive steady-state steps of a chain from a rocprofv3 kernel trace, every kernel (the runtime's copy kernels included)
with start, end, duration, queue and stream: who waits for whom across the streams of a pipelined chain.
Usage: trace_steps.py <prof_dir> [n_kernels=60]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if "ds::" in nm or "rocclr" in nm:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
SHORT = [("dcnotch", "notch"), ("binop_kernel<13", "mcspp"), ("binop_kernel<12", "mcspp12"), ("stft_rows", "rowsF"), ("subrls_fan", "rlsfan"),
         ("sublms_fan", "lmsfan"), ("fir_kernel", "fir"), ("stft_cdr", "cdr"), ("tick", "tick"), ("frames_kernel", "frames"), ("wpe", "wpe"),
         ("copyBufferRect", "copyRect"), ("copyBuffer", "copy"), ("fillBuffer", "fill")]
def short(nm):
    for k, v in SHORT:
        if k in nm:
            return v
    return nm[:40]
sel = rows[-n - 20:-20]
b = sel[0][0]
for s, e, nm, q, st in sel:
    print("%8.1f %8.1f %7.1f q%s s%s %s" % ((s - b) / 1e3, (e - b) / 1e3, (e - s) / 1e3, q, st, short(nm)))
