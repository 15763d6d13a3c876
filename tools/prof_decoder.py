"""Run the decoder N times (eager) for rocprofv3 --kernel-trace --stats / --pmc.   python tools/prof_decoder.py [M|S|Sc|L] [n]"""
import sys
import torch
from _workload import decoder_workload
name = sys.argv[1] if len(sys.argv) > 1 else 'M'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
d, x, s = decoder_workload(name)
for _ in range(n):
    d(x, s)
torch.cuda.synchronize()
