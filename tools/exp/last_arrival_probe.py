#!/usr/bin/env python3
"""Round-5 probe: can the last block of a grid read what every other block (on any XCD) wrote, with agent-scope relaxed atomics only
(no fence)?  Prints mismatch counts per mode and the time per launch."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import torch
from supervised_dispnet_amd import _lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for blocks, rec in ((208, 128), (416, 128), (1664, 128), (256, 4096), (64, 16384)):
    for mode in (0, 1):
        data = torch.zeros(blocks * rec, dtype=torch.float32, device=dev)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        result = torch.zeros(3, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rounds = 400
        e0.record()
        for r in range(rounds):
            _lib.call("dn_last_arrival_probe", data.data_ptr(), counter.data_ptr(), result.data_ptr(), blocks, rec, r + 1, mode, st)
        e1.record()
        torch.cuda.synchronize()
        res = result.cpu().tolist()
        print("blocks %5d rec %6d mode %d (%s): mismatching floats %d over %d launches (finalised %d), %.2f us per launch, counter left %d" % (
            blocks, rec, mode, "agent-scope atomics" if mode == 0 else "plain", res[0], rounds, res[2], 1e3 * e0.elapsed_time(e1) / rounds, int(counter.item())))
