"""Harness with the reference's CLI, stdout lines, -log format and pickle layout
(reference lcasr/run_dynamic_eval_full.py:31-159): load model -> per-record eval_fn -> greedy decode -> normalise ->
WER -> `WER: …` / `Average WER: …` / -log / pickle with the `_{repeat}.pkl` suffix rule.

Differences, all about where work runs:
  * the model is this package's HIP SCConformerXL; a checkpoint is {'config': {...}, 'model': state_dict} loaded with
    torch.load(weights_only=True); without -c a seeded synthetic model of the yaml config is built (no checkpoint
    exists offline);
  * eval_fn returns device log-probs (return_device=True) and the final greedy decode runs on the GPU;
  * `-kwargs chains=N` keeps N recordings in flight on each GPU (one model replica + HIP stream each, one host thread;
    same transcripts as one at a time), see lib.dynamic_eval_many;
  * launched under torchrun with N ranks, recordings shard across the GPUs (longest first) and the edit counters /
    hypotheses are gathered over RCCL at the end; rank 0 prints and saves.
Run:  python -m dynamic_asr_eval_amd.run_dynamic_eval_full -d synthetic -ds -epochs 1 -kwargs optim_lr=9e-5 …"""
import argparse
import pickle
import time

import torch

from . import dist as ddist
from . import lib
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .lib import dynamic_eval
from .model import SCConformerXL
from .tokenizer import SyntheticTokenizer, load_sentencepiece
from .wer import basic_normalize, edit_counts, rates_from_counts

normalize = basic_normalize

DEFAULT_MODEL_CONFIG = {
    'model': dict(feat_in=80, n_layers=6, d_model=768, n_heads=6, head_dim=128, subsampling_factor=8,
                  subsampling_conv_channels=256, conv_kernel_size=9, self_conditioning=True,
                  rotary_base_freq=1500000),
    'audio_chunking': {'size': 16384, 'overlap': 0},
    'training': {'max_seq_len': 0},
}


def load_model_and_tokenizer(args, device):
    """reference run_dynamic_eval_full.py:36-51 — `checkpoint['config']`, `checkpoint['model']`, strict=False."""
    if args.checkpoint:
        checkpoint = torch.load(args.checkpoint, map_location='cpu', weights_only=True)
        config, state = checkpoint['config'], checkpoint['model']
    else:
        config, state = DEFAULT_MODEL_CONFIG, None
    args.config = config
    tok_path = args.__dict__.get('tokenizer', '')
    vocab = int(args.__dict__.get('vocab_size', 4095))
    tokenizer = load_sentencepiece(tok_path) if tok_path else SyntheticTokenizer(vocab)
    model = SCConformerXL(dict(config['model']), vocab_size=tokenizer.vocab_size(), device=device)
    model.print_total_params()
    if state is not None:
        res = model.load_state_dict(state, strict=False)            # reference :47 (strict=False)
        # Parameter names are this package's (the upstream module is un-vendored): a checkpoint that matches nothing would leave
        # the zero-initialised flat buffer in place and still print a WER.  Refuse unless explicitly allowed.
        if res.missing_keys or res.unexpected_keys:
            msg = (f'checkpoint {args.checkpoint}: {len(res.missing_keys)} parameters/buffers of the model are missing '
                   f'(e.g. {res.missing_keys[:3]}), {len(res.unexpected_keys)} checkpoint keys are unknown (e.g. {res.unexpected_keys[:3]})')
            if res.missing_keys and not args.__dict__.get('allow_missing', False):
                raise KeyError(msg + '; pass `-kwargs allow_missing=True` to run with the missing tensors left at zero')
            print('WARNING: ' + msg)
        print(f'Loaded model from {args.checkpoint}')
    else:
        from .synthetic_weights import init_synthetic
        init_synthetic(model, seed=int(args.__dict__.get('seed', 0)), blank_bias=float(args.__dict__.get('blank_bias', 2.5)))
    model.device = device
    model.eval()
    return model, tokenizer


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    model, tokenizer = load_model_and_tokenizer(args, device)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.decoder.num_classes - 1, device=device)
    data = datasets_functions[args.dataset](args.split)
    if args.consistency:
        raise NotImplementedError('dynamic_eval_consistency_ctc_loss is out of scope (SURVEY.md §2 row 3)')
    eval_fn = lib.AWMC if args.awmc else dynamic_eval                   # reference run_dynamic_eval_full.py:67-72
    mine = ddist.shard_longest_first([d.get('frames', 1) for d in data], world)[rank]

    chains = int(args.__dict__.get('chains', 1))    # -kwargs chains=N: N recordings in flight per GPU (lib.dynamic_eval_many)
    lockstep = int(args.__dict__.get('lockstep', 1))   # -kwargs lockstep=R: every chain is a lockstep group of R recordings (lib._dynamic_eval_group_gen)
    models = None
    if lockstep > 1 and not args.awmc and len(mine) > 1:
        from .run_seq_eval import replicate
        models = replicate(model, max(1, min(chains, (len(mine) + lockstep - 1) // lockstep)), group=lockstep)
    elif chains > 1 and not args.awmc and len(mine) > 1:
        from .run_seq_eval import replicate
        models = replicate(model, min(chains, len(mine)))

    avg_wers = []
    for repeat in range(args.repeats):
        records = []
        if models is not None:
            loaded = [data[rec]['process_fn'](data[rec]) for rec in mine]
            stime = time.time()
            outs = lib.dynamic_eval_many(args, models, [a for a, _ in loaded], args.seq_len, args.overlap, tokenizer,
                                         use_tqdm=False, return_device=True)
            texts = [decoder(o) for o in outs]
            torch.cuda.synchronize(device)
            per_rec = (time.time() - stime) / max(1, len(mine))      # chains overlap: only the mean is meaningful
            for rec, (_, gold_text), out_text in zip(mine, loaded, texts):
                records.append({'index': rec, 'id': data[rec]['id'], 'hyp': normalize(out_text).lower(), 'gold': gold_text,
                                'elapsed': per_rec})
        for rec in (mine if models is None else []):
            if rank == 0:
                print(f'Processing {rec + 1}/{len(data)}')
                print('\n-------\n' + data[rec]['id'] + '\n-------\n')
            audio_spec, gold_text = data[rec]['process_fn'](data[rec])
            stime = time.time()
            logits = eval_fn(args, model, audio_spec, args.seq_len, args.overlap, tokenizer, beam_search_fn=None,
                             use_tqdm=(rank == 0 and not args.not_verbose), return_device=True)
            out_text = decoder(logits)
            torch.cuda.synchronize(device)
            etime = time.time()
            out = normalize(out_text).lower()
            records.append({'index': rec, 'id': data[rec]['id'], 'hyp': out, 'gold': gold_text, 'elapsed': etime - stime})
        counts = edit_counts([r['hyp'] for r in records], [r['gold'] for r in records])
        counts = ddist.all_reduce_counts(counts)           # RCCL over xGMI: 4 int64 counters
        records = ddist.gather_records(records)            # hypotheses for the pickle (small strings)
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
                    'elapsed_times': [r['elapsed'] for r in records],
                    'args_dict': {k: v for k, v in vars(args).items() if k != 'config'},
                    'repeat': f'{repeat + 1}/{args.repeats}',
                }
                save_path = args.save_path
                save_path = save_path.replace('.pkl', f'_{repeat + 1}.pkl') if save_path.endswith('.pkl') else save_path + f'_{repeat + 1}.pkl'
                with open(save_path, 'wb') as f:
                    pickle.dump(save_data, f)
                print(f'Saved to {save_path}')
        avg_wers.append(wer)
    avg = sum(avg_wers) / len(avg_wers)
    if rank == 0:
        print(f'Average WER: {avg}')
    return avg


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1, help='Number of times to repeat the evaluation')
    parser.add_argument('--save_path', '-s', type=str, default='', help='path to save')
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
