"""2-rank gloo worker for tests/test_host_cpu.py: shards 5 recordings, all-reduces the WER counters, gathers records."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamic_asr_eval_amd import dist as ddist  # noqa: E402
from dynamic_asr_eval_amd.wer import edit_counts  # noqa: E402

rank, local_rank, world = ddist.init(backend="gloo")
hyps = ["a b c", "x y", "the cat sat", "one two three four", "z"]
golds = ["a c d", "x y z", "the cat sat on the mat", "one two four", "q r"]
lengths = [30, 20, 60, 40, 10]
mine = ddist.shard_longest_first(lengths, world)[rank]
counts = edit_counts([hyps[i] for i in mine], [golds[i] for i in mine])
total = ddist.all_reduce_counts(counts)
records = ddist.gather_records([{"index": i, "id": f"rec{i}"} for i in mine])
mx = ddist.max_over_ranks(1.0 + rank)
ddist.barrier()
if rank == 0:
    print("RESULT " + json.dumps({"counts": list(total), "expected_counts": list(edit_counts(hyps, golds)),
                                  "ids": [r["id"] for r in records], "max_elapsed": mx}), flush=True)
