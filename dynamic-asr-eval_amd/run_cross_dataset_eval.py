"""Cross-dataset adaptation harness (BASELINE config 5) with the reference's flow, flags and pickle layout
(reference lcasr/run_cross_dataset_eval.py:32-218): baselines on A and B with epochs=0 (:92-142), then for every i in A:
adapt on A[i] with return_params=True (:147-156), load the adapted weights (:157-158), evaluate all of B (:160-176) and
A minus {i} (:178-195), restore (:197-198); results {a_baseline, b_baseline, a_to_b, a_to_a_loo, ...} (:200-218).

Multi-GPU: the outer iterations i are independent (weights are restored between them), so ranks take
i in A longest-first; the baselines shard by recording and their edit counters are all-reduced over RCCL; the per-i result
lists are gathered to rank 0 in i order.  One process per GPU, no data-path collective."""
import argparse
import pickle

import torch

from . import dist as ddist
from . import lib
from .decoding import GreedyCTCDecoder
from .datasets import datasets_functions
from .harness_common import clone_params, restore_params, score_texts, set_params, transcribe
from .run_dynamic_eval_full import load_model_and_tokenizer


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    model, tokenizer = load_model_and_tokenizer(args, device)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.decoder.num_classes - 1, device=device)
    data_a = datasets_functions[args.dataset](args.split)
    data_b = datasets_functions[args.dataset2](args.split)
    if rank == 0:
        print(f'Dataset A ({args.dataset}): {len(data_a)} records')
        print(f'Dataset B ({args.dataset2}): {len(data_b)} records')
    eval_fn = lib.dynamic_eval if not args.awmc else lib.AWMC
    adapt_overlap = args.adapt_overlap if args.adapt_overlap is not None else args.overlap
    original = clone_params(model)
    args_dict = vars(args).copy()
    args_dict['epochs'] = 0
    baseline_args = argparse.Namespace(**args_dict)

    def evaluate(records, idxs):
        golds, preds = [], []
        for j in idxs:
            audio_spec, gold_text = records[j]['process_fn'](records[j])
            logits = eval_fn(baseline_args, model, audio_spec, args.seq_len, args.overlap, tokenizer, beam_search_fn=None,
                             use_tqdm=False, return_device=True)
            preds.append(transcribe(decoder, logits))
            golds.append(gold_text)
        return preds, golds

    for repeat in range(args.repeats):
        if rank == 0:
            print(f'\n=== Repeat {repeat + 1}/{args.repeats} ===')
        shard_a = ddist.shard_longest_first([d.get('frames', 1) for d in data_a], world)[rank]
        shard_b = ddist.shard_longest_first([d.get('frames', 1) for d in data_b], world)[rank]
        a_baseline = score_texts(*evaluate(data_a, shard_a), reduce_over_ranks=True)
        b_baseline = score_texts(*evaluate(data_b, shard_b), reduce_over_ranks=True)
        if rank == 0:
            print(f'A baseline WER: {a_baseline["wer"]}')
            print(f'B baseline WER: {b_baseline["wer"]}')
        mine = []
        for i in shard_a:                                           # A-X: adapt on A[i], eval on B and A \ {i}
            audio_spec, _ = data_a[i]['process_fn'](data_a[i])
            _, updated = eval_fn(args, model, audio_spec, args.seq_len, adapt_overlap, tokenizer, beam_search_fn=None,
                                 use_tqdm=False, return_params=True, return_device=True)
            set_params(model, updated)
            ab = score_texts(*evaluate(data_b, range(len(data_b))))
            aa = score_texts(*evaluate(data_a, [k for k in range(len(data_a)) if k != i]))
            mine.append({'index': i, 'a_to_b': ab, 'a_to_a_loo': aa})
            restore_params(model, original)
        allr = ddist.gather_records(mine)
        if rank == 0:
            results = {'a_baseline': a_baseline, 'b_baseline': b_baseline, 'a_to_b': [r['a_to_b'] for r in allr],
                       'a_to_a_loo': [r['a_to_a_loo'] for r in allr], 'dataset_a': args.dataset, 'dataset_b': args.dataset2,
                       'args_dict': {k: v for k, v in vars(args).items() if k != 'config'}, 'repeat': f'{repeat + 1}/{args.repeats}'}
            if args.save_path != '':
                sp = args.save_path
                sp = sp.replace('.pkl', f'_{repeat + 1}.pkl') if sp.endswith('.pkl') else sp + f'_{repeat + 1}.pkl'
                with open(sp, 'wb') as f:
                    pickle.dump(results, f)
                print(f'Finished and saved to {sp}')
    return 0


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--dataset2', '-d2', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1)
    parser.add_argument('--save_path', '-s', type=str, default='')
    parser.add_argument('--adapt_overlap', '-ao', type=int, default=None)
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
