"""One of bench.py's side workloads alone (twice, best of two), so that `rocprofv3 --kernel-trace --stats -- python3 scripts/probe_side_workload.py
<awmc|wav2vec2_su|enc_dec_teacher_ce>` shows where that loop's time goes.  Prints the same record bench.py puts under `other_workloads`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.synthetic_weights import init_synthetic

which = sys.argv[1]
sys.argv = sys.argv[:1]
a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = None
if which == 'awmc':
    model = SCConformerXL(vocab_size=a.vocab, device=dev)
    init_synthetic(model, seed=0, blank_bias=0.0)
print(json.dumps(bench.other_workloads(a, model, dev, which=(which,))))
