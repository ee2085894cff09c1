"""One of bench.py's side measurements on its own (python scripts/bench_side.py wav2vec2_su [awmc ...]): prints the `other_workloads` entries as JSON."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

which = tuple(sys.argv[1:]) or ('wav2vec2_su',)
sys.argv = sys.argv[:1]
a = bench.parse()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
model = None
if any(w in which for w in ('awmc',)):
    raise SystemExit("awmc needs the benchmark's model: run bench.py")
print(json.dumps(bench.other_workloads(a, model, dev, which=which)))
