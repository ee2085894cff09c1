"""Whole-concat adaptation harness (BASELINE config 4) with the reference's flow (reference
lcasr/run_whole_concat_eval.py:43-190): concatenate every record's spectrogram along time (:121-127), per repeat: baseline
eval of every record with epochs=0 (:138), `adapt_on_concat_only` on the concatenation (:141-148), load the adapted weights
(:149-150), re-evaluate every record (:152), report Baseline / Adapted / Delta WER, pickle (:157-183), restore (:186).

Multi-GPU: the adaptation over the concatenated stream is ONE sequential chain, so it runs as
replicas (rank 0 adapts, the adapted flat parameter buffer — ~0.36 GB — is broadcast once over RCCL/xGMI); the before/after
per-record evaluations shard by recording and their edit counters are all-reduced."""
import argparse
import copy
import pickle

import torch
import torch.distributed as tdist

from . import dist as ddist
from . import lib
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .harness_common import clone_params, restore_params, score_texts, set_params, transcribe
from .run_dynamic_eval_full import load_model_and_tokenizer
from .run_half_concat_eval import adapt_on_concat_only, concatenate_specs


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    model, tokenizer = load_model_and_tokenizer(args, device)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.decoder.num_classes - 1, device=device)
    data = datasets_functions[args.dataset](args.split)
    eval_fn = lib.dynamic_eval if not args.awmc else lib.AWMC
    adapt_overlap = args.adapt_overlap if args.adapt_overlap is not None else args.overlap
    original = clone_params(model)
    baseline_args = copy.copy(args)
    baseline_args.epochs = 0
    shard = ddist.shard_longest_first([d.get('frames', 1) for d in data], world)[rank]

    def evaluate_records(eval_args):
        preds, golds, per_record = [], [], []
        for j in shard:
            rec = data[j]
            audio_spec, gold_text = rec['process_fn'](rec)
            logits = eval_fn(eval_args, model, audio_spec, args.seq_len, args.overlap, tokenizer, use_tqdm=False, beam_search_fn=None,
                             return_device=True)
            pred = transcribe(decoder, logits)
            preds.append(pred); golds.append(gold_text)
            per_record.append({'index': j, 'id': rec['id'], 'prediction': pred, 'gold': gold_text})
        return score_texts(preds, golds, reduce_over_ranks=True), ddist.gather_records(per_record)

    concat_spec = concatenate_specs([rec['process_fn'](rec)[0].to(device) for rec in data]) if rank == 0 else None
    all_ids = [rec['id'] for rec in data]
    scores = []
    for repeat in range(args.repeats):
        restore_params(model, original)
        baseline_scores, baseline_per_record = evaluate_records(baseline_args)
        if rank == 0:
            print(f'Baseline WER = {baseline_scores["wer"]}')
            updated = adapt_on_concat_only(args, model, concat_spec, tokenizer, beamsearch=None, adapt_overlap=adapt_overlap)
            set_params(model, updated)
        if world > 1:                                               # adapted weights to every replica: one broadcast
            tdist.broadcast(model.flat_params, src=0)
        adapted_scores, adapted_per_record = evaluate_records(baseline_args)
        if rank == 0:
            print(f'Adapted WER = {adapted_scores["wer"]}')
            print(f'Delta = {adapted_scores["wer"] - baseline_scores["wer"]:+.6f}')
            res = {'dataset': args.dataset, 'split': args.split, 'repeat': f'{repeat + 1}/{args.repeats}', 'adapt_ids': all_ids,
                   'adapt_num_records': len(all_ids), 'concat_spec_shape': tuple(concat_spec.shape),
                   'concat_total_frames': int(concat_spec.shape[-1]), 'baseline': baseline_scores, 'adapted': adapted_scores,
                   'delta_wer': adapted_scores['wer'] - baseline_scores['wer'],
                   'baseline_model_output': [r['prediction'] for r in baseline_per_record],
                   'model_output': [r['prediction'] for r in adapted_per_record], 'gold': [r['gold'] for r in adapted_per_record],
                   'baseline_per_record': baseline_per_record, 'adapted_per_record': adapted_per_record,
                   'args_dict': {k: v for k, v in vars(args).items() if k != 'config'}}
            if args.save_path != '':
                sp = args.save_path
                sp = sp.replace('.pkl', f'_{repeat + 1}.pkl') if sp.endswith('.pkl') else sp + f'_{repeat + 1}.pkl'
                with open(sp, 'wb') as f:
                    pickle.dump(res, f)
                print(f'Saved to {sp}')
        scores.append(adapted_scores['wer'])
        restore_params(model, original)
    if scores and rank == 0:
        print(f'Average adapted WER across repeats: {sum(scores) / len(scores)}')
    return sum(scores) / len(scores) if scores else None


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1)
    parser.add_argument('--save_path', '-s', type=str, default='')
    parser.add_argument('--adapt_overlap', '-ao', type=int, default=None)
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
