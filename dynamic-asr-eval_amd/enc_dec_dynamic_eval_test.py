"""Harness mirror of the reference's lcasr/enc_dec_dynamic_eval_test.py:34-154 — same flags (`--dataset/-d`, `--repeats/-r`,
`--save_path/-s`, `--breaks`, `--training_mode`, `--maxrl_success_threshold`, `--grpo_normalize_std`, the teacher-filter flags, plus
`lib.apply_args`), same stdout lines (`Processing i/n`, `WER:`, `Average WER:`, `Saved to`), same `-log` line and pickle keys
(:102-123).  Differences: the model is this package's EncDecSCConformerXL (checkpoint = {'config', 'model'} loaded with
torch.load(weights_only=True), or seeded weights without -c); only `--training_mode teacher_ce` is implemented (the default
`grpo` and `maxrl` raise: RL modes are out of scope); datasets are the synthetic adapters."""
import argparse
import pickle
import time

import torch

from . import lib
from .datasets import datasets_functions
from .enc_dec import DEFAULT_DECODER, EncDecSCConformerXL, enc_dec_dynamic_eval
from .enc_dec_teacher_filters import add_enc_dec_teacher_filter_args
from .run_dynamic_eval_full import DEFAULT_MODEL_CONFIG
from .tokenizer import SyntheticTokenizer, load_sentencepiece
from .wer import basic_normalize as normalize, word_error_rate_detail


def load_enc_dec_model(args, device):
    """reference enc_dec_dynamic_eval_test.py:38-52 (`checkpoint['config']`, `checkpoint['model']`, strict=False)."""
    if args.checkpoint:
        checkpoint = torch.load(args.checkpoint, map_location='cpu', weights_only=True)
        config, state = checkpoint['config'], checkpoint['model']
    else:
        config, state = DEFAULT_MODEL_CONFIG, None
    args.config = config
    tok_path = args.__dict__.get('tokenizer', '')
    tokenizer = load_sentencepiece(tok_path) if tok_path else SyntheticTokenizer(int(args.__dict__.get('vocab_size', 4095)))
    model = EncDecSCConformerXL(dict(config['model']), vocab_size=tokenizer.vocab_size(), device=device)
    model.print_total_params()
    if state is not None:
        res = model.load_state_dict(state, strict=False)
        if res.missing_keys and not args.__dict__.get('allow_missing', False):
            raise KeyError(f'checkpoint {args.checkpoint}: {len(res.missing_keys)} parameters of the model are missing (e.g. {res.missing_keys[:3]}); '
                           'pass `-kwargs allow_missing=True` to run with them left at zero')
        print(f'Loaded model from {args.checkpoint}')
    else:
        from .synthetic_weights import init_synthetic
        init_synthetic(model, seed=int(args.__dict__.get('seed', 0)), blank_bias=float(args.__dict__.get('blank_bias', 2.5)))
    model.device = device
    model.eval()
    return model, tokenizer


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    device = torch.device('cuda', 0)
    model, tokenizer = load_enc_dec_model(args, device)
    data = datasets_functions[args.dataset](args.split)
    avg_wers = []
    for repeat in range(args.repeats):
        all_texts, all_golds, elapsed_times = [], [], []
        for rec in range(len(data)):
            print(f'Processing {rec + 1}/{len(data)}')
            print('\n-------\n' + data[rec]['id'] + '\n-------\n')
            audio_spec, gold_text = data[rec]['process_fn'](data[rec])
            stime = time.time()
            model_out = enc_dec_dynamic_eval(args=args, model=model, spec=audio_spec, seq_len=args.seq_len, overlap=0, tokenizer=tokenizer,
                                             use_tqdm=not args.not_verbose)
            torch.cuda.synchronize(device)
            elapsed_times.append(time.time() - stime)
            out = normalize(model_out).lower()
            if not args.not_verbose:
                print(gold_text, '\n', out, '\n\n')
            all_texts.append(out)
            all_golds.append(gold_text)
            if args.breaks:
                break
        wer, words, ins_rate, del_rate, sub_rate = word_error_rate_detail(hypotheses=all_texts, references=all_golds)
        print(f'WER: {wer}')
        if args.log != '':
            with open(args.log, 'a') as f:
                f.write(f'{args.checkpoint}\t overlap: {args.overlap}\t seq_len: {args.seq_len}\t WER: {wer}\n')
        if args.save_path != '':
            save_data = {'wer': wer, 'words': words, 'ins_rate': ins_rate, 'del_rate': del_rate, 'sub_rate': sub_rate,
                         'model_output': all_texts, 'gold': all_golds, 'elapsed_times': elapsed_times,
                         'args_dict': {k: v for k, v in vars(args).items() if k != 'config'}, 'repeat': f'{repeat + 1}/{args.repeats}'}
            save_path = args.save_path
            save_path = save_path.replace('.pkl', f'_{repeat + 1}.pkl') if save_path.endswith('.pkl') else save_path + f'_{repeat + 1}.pkl'
            with open(save_path, 'wb') as f:
                pickle.dump(save_data, f)
            print(f'Saved to {save_path}')
        avg_wers.append(wer)
    avg = sum(avg_wers) / len(avg_wers)
    print(f'Average WER: {avg}')
    return avg


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1, help='Number of times to repeat the evaluation')
    parser.add_argument('--save_path', '-s', type=str, default='', help='path to save')
    parser.add_argument('--breaks', action='store_true', help='Break after first sample (for debugging)')
    parser.add_argument('--training_mode', type=str, default='grpo', choices=['grpo', 'maxrl', 'teacher_ce'])
    parser.add_argument('--maxrl_success_threshold', type=float, default=0.9)
    parser.add_argument('--grpo_normalize_std', action=argparse.BooleanOptionalAction, default=True)
    add_enc_dec_teacher_filter_args(parser)
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
