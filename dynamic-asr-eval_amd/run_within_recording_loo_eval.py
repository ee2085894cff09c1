"""Within-recording leave-one-out harness with the reference's flow, flags and outputs
(reference lcasr/run_within_recording_loo_eval.py:40-238): per recording, a no-adapt windowed baseline (:197-199), then
the recording is cut into OUTER chunks (`-loo_s / -loo_o`, :104); for every chunk i the model is adapted on chunk i
(eval_fn with return_params, :133-146), the adapted weights are loaded (:147-148) and every chunk j whose audio is
DISJOINT from i (:116-119) is transcribed with epochs=0 (:150-157); probabilities are accumulated at row
j // downsampling_factor (:155-157), rows nobody covered are dropped (:161-175), log(sum / count) is decoded (:177-181).
Fallbacks for <= 1 chunk or no disjoint pair: the windowed baseline (:106-126).

Where the work runs here: accumulators in HBM (dyn_stitch_accumulate at an explicit row, dyn_stitch_finalize_rows for
coverage with gaps); adapted weights are kept as one device-to-device copy of the flat parameter buffer instead of CPU
clones; the epochs=0 inferences of one adapted model are independent, so `-kwargs chains=N` runs N at a time."""
import argparse
import copy
import pickle

import torch

from . import dist as ddist
from . import lib, ops
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .harness_common import clone_params, normalize, restore_params, set_params
from .lib import AWMC, dynamic_eval, prepare_chunks
from .run_dynamic_eval_full import load_model_and_tokenizer
from .run_seq_eval import replicate
from .wer import edit_counts, rates_from_counts


def disjoint_pairs(chunk_keys, chunk_len):
    """reference :116-121 — chunk k covers frames [k, k + len_k)."""
    def disjoint(ai, ej):
        return ej >= ai + chunk_len[ai] or ai >= ej + chunk_len[ej]
    return {ai: [ej for ej in chunk_keys if disjoint(ai, ej)] for ai in chunk_keys}


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    model, tokenizer = load_model_and_tokenizer(args, device)
    num_classes = model.decoder.num_classes
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=num_classes - 1, device=device)
    data = datasets_functions[args.dataset](args.split)
    eval_fn = dynamic_eval if not args.awmc else AWMC
    original = clone_params(model)
    downsampling_factor = args.config['model']['subsampling_factor']
    baseline_args = copy.copy(args)
    baseline_args.epochs = 0
    chains = int(args.__dict__.get('chains', 1))
    models = replicate(model, chains) if (chains > 1 and not args.awmc) else [model]

    def windowed_inference(chunks):
        """epochs=0 inference of several chunks with the CURRENT weights of `model` -> device log-probs."""
        if len(models) > 1 and len(chunks) > 1:
            for m in models[1:]:
                m.flat_params.copy_(model.flat_params)
            return lib.dynamic_eval_many(baseline_args, models, chunks, args.seq_len, args.overlap, tokenizer, use_tqdm=False,
                                         return_device=True)
        return [eval_fn(baseline_args, model, c, args.seq_len, args.overlap, tokenizer, use_tqdm=False, beam_search_fn=None,
                        return_device=True) for c in chunks]

    def transcribe(logits):
        return normalize(decoder(logits)).lower()

    def loo_eval(audio_dev):
        spec_n = audio_dev.shape[-1]
        chunks, chunk_keys = prepare_chunks(audio_dev, args.loo_seq_len, args.loo_overlap)
        chunk_keys = sorted(chunk_keys)
        n_chunks = len(chunk_keys)
        if n_chunks <= 1:
            print(f'  Only {n_chunks} LOO chunk(s) at loo_seq_len={args.loo_seq_len}; falling back to windowed no-adapt eval on full recording.')
            return windowed_inference([audio_dev])[0], {'n_chunks': n_chunks, 'mode': 'fallback_windowed_eval'}
        chunk_len = {k: chunks[k].shape[-1] for k in chunk_keys}
        valid_evals = disjoint_pairs(chunk_keys, chunk_len)
        valid_pairs = sum(len(v) for v in valid_evals.values())
        if valid_pairs == 0:
            print(f'  {n_chunks} LOO chunks but no audio-disjoint (i, j) pairs at loo_seq_len={args.loo_seq_len}; falling back to windowed no-adapt eval on full recording.')
            return windowed_inference([audio_dev])[0], {'n_chunks': n_chunks, 'mode': 'fallback_no_disjoint_pairs'}
        usable_adapts = [ai for ai in chunk_keys if valid_evals[ai]]
        print(f'  {n_chunks} LOO chunks -> {len(usable_adapts)} adaptations + {valid_pairs} windowed inferences (audio-disjoint LOO)')

        rows = spec_n // downsampling_factor + args.loo_seq_len
        acc = torch.zeros(rows, num_classes, device=device, dtype=torch.float32)
        cnt = torch.zeros(rows, device=device, dtype=torch.float32)
        for adapt_i in usable_adapts:
            restore_params(model, original)
            _, updated = eval_fn(args, model, chunks[adapt_i], args.seq_len, args.overlap, tokenizer, use_tqdm=False,
                                 beam_search_fn=None, return_params=True, return_device=True)
            set_params(model, updated)
            evals = valid_evals[adapt_i]
            for eval_j, lp in zip(evals, windowed_inference([chunks[j] for j in evals])):
                ops.stitch_accumulate(lp, acc, cnt, eval_j // downsampling_factor)
        restore_params(model, original)

        covered = torch.nonzero(cnt > 0).squeeze(-1)           # index plumbing; the arithmetic is dyn_stitch_finalize_rows
        if covered.numel() == 0:
            raise RuntimeError('LOO stitching produced no coverage at any position.')
        first, last = int(covered[0]), int(covered[-1])
        if covered.numel() != last - first + 1:
            gap = last - first + 1 - covered.numel()
            print(f'  WARNING: audio-disjoint LOO stitching has {gap} uncovered position(s) inside covered span [{first}, {last}]; uncovered positions are dropped before decoding.')
        return ops.stitch_finalize_rows(acc, cnt, covered.contiguous()), {'n_chunks': n_chunks, 'mode': 'loo'}

    mine = ddist.shard_longest_first([d.get('frames', 1) for d in data], world)[rank]
    loo_wers, base_wers = [], []
    for repeat in range(args.repeats):
        if rank == 0:
            print(f'\n=== Repeat {repeat + 1}/{args.repeats} ===')
        records = []
        for rec_idx in mine:
            rec = data[rec_idx]
            print(f'\n-------\n{rec["id"]}\n-------')
            audio_spec, gold_text = rec['process_fn'](rec)
            audio_dev = audio_spec.to(device=device, dtype=torch.float32)
            restore_params(model, original)
            baseline_pred = transcribe(windowed_inference([audio_dev])[0])
            print(f'BASELINE: {baseline_pred}')
            stitched, info = loo_eval(audio_dev)
            pred = transcribe(stitched)
            print(f'GOLD:     {gold_text}')
            print(f'LOO PRED: {pred}')
            records.append({'index': rec_idx, 'id': rec['id'], 'hyp': pred, 'baseline': baseline_pred, 'gold': gold_text, 'meta': info})
        loo_counts = ddist.all_reduce_counts(edit_counts([r['hyp'] for r in records], [r['gold'] for r in records]))
        base_counts = ddist.all_reduce_counts(edit_counts([r['baseline'] for r in records], [r['gold'] for r in records]))
        records = ddist.gather_records(records)
        def scores(counts):
            wer, words, ins, dele, sub = rates_from_counts(*counts)
            return {'wer': wer, 'words': words, 'ins_rate': ins, 'del_rate': dele, 'sub_rate': sub}

        loo, base = scores(loo_counts), scores(base_counts)
        if rank == 0:
            print(f'\nRepeat {repeat + 1} baseline WER: {base["wer"]}')           # reference :213-215
            print(f'Repeat {repeat + 1} LOO WER:      {loo["wer"]}')
            print(f'Repeat {repeat + 1} delta:        {loo["wer"] - base["wer"]:+.4f}')
            if args.save_path != '':
                save_data = {                                                        # reference :218-228
                    'loo': loo, 'baseline': base,
                    'model_output': [r['hyp'] for r in records], 'baseline_model_output': [r['baseline'] for r in records],
                    'gold': [r['gold'] for r in records],
                    'per_recording_meta': [{'id': r['id'], **r['meta']} for r in records],
                    'dataset': args.dataset,
                    'args_dict': {k: v for k, v in vars(args).items() if k != 'config'},
                    'repeat': f'{repeat + 1}/{args.repeats}',
                }
                save_path = args.save_path
                save_path = save_path.replace('.pkl', f'_{repeat + 1}.pkl') if save_path.endswith('.pkl') else save_path + f'_{repeat + 1}.pkl'
                with open(save_path, 'wb') as f:
                    pickle.dump(save_data, f)
                print(f'Saved to {save_path}')
        loo_wers.append(loo['wer'])
        base_wers.append(base['wer'])
    return sum(loo_wers) / len(loo_wers)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1, help='Number of times to repeat the evaluation')
    parser.add_argument('--save_path', '-s', type=str, default='', help='path to save')
    parser.add_argument('--loo_seq_len', '-loo_s', type=int, default=65536, help='Outer LOO chunk length (over which we iterate leave-one-out).')
    parser.add_argument('--loo_overlap', '-loo_o', type=int, default=57344, help='Outer LOO chunk overlap (stride = loo_seq_len - loo_overlap).')
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
