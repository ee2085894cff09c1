"""In-dataset transfer harness with the reference's flow, flags and outputs (reference lcasr/run_in_dataset_eval.py:32-167):
adapt on the FIRST recording of the split (eval_fn with return_params, :79-91; `-ao/--adapt_overlap` overrides the overlap of
that pass only, :73-75), load the adapted weights into the model (:93-94), then transcribe every other recording with
epochs = 0 (:98-118), greedy decode, normalise, WER, -log line, `_{repeat}.pkl` pickle (:136-160).

Kept as in the reference: the adapted weights stay loaded, so with `-r N` every repeat adapts on top of the previous repeat's
weights (nothing restores them between repeats, :77-94).  Here the adapted parameters stay in HBM (one device-to-device copy of
the flat buffer), the epochs = 0 evaluations shard across ranks (each rank adapts on recording 0 itself: the chain is sequential
and cheap next to the evaluations) and `-kwargs chains=N` runs N of them at a time on each GPU."""
import argparse
import pickle

import torch

from . import dist as ddist
from . import lib
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .harness_common import normalize, set_params
from .lib import AWMC, dynamic_eval
from .run_dynamic_eval_full import load_model_and_tokenizer
from .run_seq_eval import replicate
from .wer import edit_counts, rates_from_counts


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    model, tokenizer = load_model_and_tokenizer(args, device)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.decoder.num_classes - 1, device=device)
    data = datasets_functions[args.dataset](args.split)
    eval_fn = dynamic_eval if not args.awmc else AWMC
    adapt_overlap = args.adapt_overlap if args.adapt_overlap is not None else args.overlap
    if adapt_overlap != args.overlap and rank == 0:
        print(f'Using adapt_overlap={adapt_overlap} for adaptation (eval overlap={args.overlap})')
    chains = int(args.__dict__.get('chains', 1))
    rest = data[1:]
    mine = ddist.shard_longest_first([d.get('frames', 1) for d in rest], world)[rank]
    args_dict = vars(args).copy()
    args_dict['epochs'] = 0
    d2_args = argparse.Namespace(**args_dict)

    wers = []
    for repeat in range(args.repeats):
        audio_spec, _ = data[0]['process_fn'](data[0])
        _, updated = eval_fn(args, model, audio_spec, args.seq_len, adapt_overlap, tokenizer, beam_search_fn=None, use_tqdm=False,
                             return_params=True, return_device=True)
        set_params(model, updated)                                     # reference :93-94
        records = []
        loaded = [rest[j]['process_fn'](rest[j]) for j in mine]
        if chains > 1 and not args.awmc and len(mine) > 1:
            models = replicate(model, min(chains, len(mine)))
            outs = lib.dynamic_eval_many(d2_args, models, [a for a, _ in loaded], args.seq_len, args.overlap, tokenizer, use_tqdm=False,
                                         return_device=True)
        else:
            outs = [eval_fn(d2_args, model, a, args.seq_len, args.overlap, tokenizer, beam_search_fn=None, use_tqdm=False,
                            return_device=True) for a, _ in loaded]
        for j, (_, gold_text), logits in zip(mine, loaded, outs):
            out = normalize(decoder(logits)).lower()
            if rank == 0 and not args.not_verbose:
                print(f'Processing {j + 1}/{len(rest)}')
                print('\n-------\n' + rest[j]['id'] + '\n-------\n')
                print(gold_text, '\n', out, '\n\n')
            records.append({'index': j, 'id': rest[j]['id'], 'hyp': out, 'gold': gold_text})
        counts = ddist.all_reduce_counts(edit_counts([r['hyp'] for r in records], [r['gold'] for r in records]))
        records = ddist.gather_records(records)
        wer, words, ins_rate, del_rate, sub_rate = rates_from_counts(*counts)
        if rank == 0:
            print(f'WER: {wer}')
            if args.log != '':
                with open(args.log, 'a') as f:
                    f.write(f'{args.checkpoint}\t overlap: {args.overlap}\t seq_len: {args.seq_len}\t WER: {wer}\n')
            if args.save_path != '':
                save_data = {
                    'wer': wer, 'words': words, 'ins_rate': ins_rate, 'del_rate': del_rate, 'sub_rate': sub_rate,
                    'model_output': [r['hyp'] for r in records], 'gold': [r['gold'] for r in records],
                    'args_dict': {k: v for k, v in vars(args).items() if k != 'config'},
                    'repeat': f'{repeat + 1}/{args.repeats}',
                }
                save_path = args.save_path
                save_path = save_path.replace('.pkl', f'_{repeat + 1}.pkl') if save_path.endswith('.pkl') else save_path + f'_{repeat + 1}.pkl'
                with open(save_path, 'wb') as f:
                    pickle.dump(save_data, f)
        wers.append(wer)
    avg = sum(wers) / len(wers)
    if rank == 0:
        print(f'Average WER: {avg}')
    return avg


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1, help='Number of times to repeat the evaluation')
    parser.add_argument('--save_path', '-s', type=str, default='', help='path to save')
    parser.add_argument('--adapt_overlap', '-ao', type=int, default=None,
                        help='Overlap used during adaptation passes only. If unset, adaptation uses --overlap (current behavior).')
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
