"""Harness mirror of the reference's lcasr/enc_dec_inference_test.py:33-113 (BASELINE config 1's named driver): load the model,
decode every record with `enc_dec_inference` (windows of `-seq` frames, no overlap, greedy), normalise, WER, `-log` line, pickle with
`repeat: '1/1'` (:92-105).  `--decoding_mode joint` (CTC + LM beam search through pyctcdecode, un-vendored) is out of scope and raises;
`--ctc_greedy` additionally prints the CTC-head greedy transcript of the same windows (the "CTC greedy" plumbing case of config 1)."""
import argparse
import pickle

import torch

from . import lib
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .enc_dec import enc_dec_inference
from .enc_dec_dynamic_eval_test import load_enc_dec_model
from .lib import prepare_chunks
from .wer import basic_normalize as normalize, word_error_rate_detail


def ctc_greedy_inference(model, spec, seq_len, tokenizer):
    """CTC-head greedy decode over the same non-overlapping windows (text per window joined by spaces)."""
    dec = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.ctc_decoder.num_classes - 1, device=model.device)
    spec = spec.to(device=model.device, dtype=torch.float32)
    data, keys = prepare_chunks(spec, seq_len, 0)
    texts = []
    for k in keys:
        with torch.no_grad():
            texts.append(dec(model.forward(data[k].contiguous())['final_posteriors_ctc'][0]).strip())
    return " ".join(texts).replace('  ', ' ').strip()


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    if args.decoding_mode != 'default':
        raise NotImplementedError("decoding_mode 'joint' (enc_dec_ctc_beamsearch_inference, pyctcdecode) is out of scope")
    device = torch.device('cuda', 0)
    model, tokenizer = load_enc_dec_model(args, device)
    data = datasets_functions[args.dataset](args.split)
    all_texts, all_golds = [], []
    for rec in range(len(data)):
        print(f'Processing {rec + 1}/{len(data)}')
        print('\n-------\n' + data[rec]['id'] + '\n-------\n')
        audio_spec, gold_text = data[rec]['process_fn'](data[rec])
        model_out = enc_dec_inference(model=model, spec=audio_spec, seq_len=args.seq_len, overlap=0, tokenizer=tokenizer,
                                      use_tqdm=not args.not_verbose)
        out = normalize(model_out).lower()
        if args.ctc_greedy:
            print('CTC greedy:', normalize(ctc_greedy_inference(model, audio_spec, args.seq_len, tokenizer)).lower())
        if not args.not_verbose:
            print(gold_text, '\n', out, '\n\n')
        all_texts.append(out)
        all_golds.append(gold_text)
    wer, words, ins_rate, del_rate, sub_rate = word_error_rate_detail(hypotheses=all_texts, references=all_golds)
    print(f'WER: {wer}')
    if args.log != '':
        with open(args.log, 'a') as f:
            f.write(f'{args.checkpoint}\t overlap: {args.overlap}\t seq_len: {args.seq_len}\t WER: {wer}\n')
    if args.save_path != '':
        save_data = {'wer': wer, 'words': words, 'ins_rate': ins_rate, 'del_rate': del_rate, 'sub_rate': sub_rate, 'model_output': all_texts,
                     'gold': all_golds, 'args_dict': {k: v for k, v in vars(args).items() if k != 'config'}, 'repeat': f'{1}/{1}'}
        save_path = args.save_path
        save_path = save_path.replace('.pkl', f'_{1}.pkl') if save_path.endswith('.pkl') else save_path + f'_{1}.pkl'
        with open(save_path, 'wb') as f:
            pickle.dump(save_data, f)
    return wer


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', '-d', type=str, default='synthetic', choices=datasets_functions.keys())
    parser.add_argument('-mode', '--decoding_mode', type=str, default='default', choices=['default', 'joint'])
    parser.add_argument('--save_path', '-s', type=str, default='', help='path to save')
    parser.add_argument('-alpha', type=float, default=0.816, help='LM weight')
    parser.add_argument('-beta', type=float, default=1.11, help='non-blank bonus')
    parser.add_argument('--ctc_greedy', action='store_true', help='also print the CTC-head greedy transcript')
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
