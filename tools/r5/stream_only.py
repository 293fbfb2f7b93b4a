#!/usr/bin/env python3
"""GPU box: bench.py's two streaming entries alone (default batch and 256 rows per launch) with this run's PCIe bound."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
pcie = bench.pcie_copy_rates(torch, dev)
for name, batch in (("streaming", 0), ("streaming_batch256", 256)):
    print(json.dumps({name: bench.streaming_leg(2.5, batch, pcie)}), flush=True)
