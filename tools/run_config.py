#!/usr/bin/env python3
"""One of bench.py's secondary configurations on its own (for `rocprofv3 --kernel-trace --stats -- python3 tools/run_config.py 2`)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(bench.secondary_configs(dev, 20, steps=steps, only="configs[%s]" % which, kernels=False)))
