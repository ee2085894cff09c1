"""Outer-window ("NSTI over long sequences") harness with the reference's flow, flags and outputs
(reference lcasr/run_seq_eval.py:36-196): each recording is cut into OUTER windows of `-nsti_s` frames with `-nsti_o`
overlap (prepare_chunks, :105); every outer window goes through eval_fn with the inner `-seq / -o` (:110-119), and the
outer windows' posteriors are stitched with the same exp / overlap-add / count / log rule (:120-142) before the greedy
decode (:146-149), normalise, WER, `-log` line and pickle (:161-190).  `epochs == 0` means one outer window = the whole
recording, no overlap (:98-100).

Where the work runs here: the outer stitch accumulators live in HBM (dyn_stitch_accumulate / dyn_stitch_finalize; the
reference builds two [spec_n//4 + seq_len, V+1] host buffers per recording and ships every outer window's posteriors
over PCIe, :102,120), eval_fn returns device log-probs, and since outer windows are independent (weights are restored by
every eval_fn call, lib.py:636-637) `-kwargs chains=N` keeps N of them in flight on the GPU (lib.dynamic_eval_many).
The reference keeps only recordings of >= 60 min (ffmpeg.probe, :63); here `min_minutes` (default 60, -kwargs) filters
on the adapter's frame count.  Recordings shard across ranks like run_dynamic_eval_full."""
import argparse
import pickle

import torch

from . import dist as ddist
from . import lib, ops
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .harness_common import normalize
from .lib import AWMC, dynamic_eval, prepare_chunks
from .model import SCConformerXL
from .run_dynamic_eval_full import load_model_and_tokenizer
from .wer import edit_counts, rates_from_counts


def outer_stitch(windows, overlap, num_classes, device):
    """windows: [(key, log_probs [ds_len, C] CUDA, u_len)] -> stitched log-probs [T_ds, C] on device.
    reference run_seq_eval.py:120-142 (logit_position -= overlap_ds except for key 0; sum / count; log)."""
    total = sum(lp.shape[0] for _, lp, _ in windows)
    acc = torch.zeros(total, num_classes, device=device, dtype=torch.float32)
    cnt = torch.zeros(total, device=device, dtype=torch.float32)
    pos = end = 0
    for key, lp, u_len in sorted(windows, key=lambda w: w[0]):
        ds_len = lp.shape[0]
        overlap_ds = int(overlap / (u_len / ds_len))
        pos -= overlap_ds if key != 0 else 0
        ops.stitch_accumulate(lp, acc, cnt, pos)
        pos += ds_len
        end = max(end, pos)
    return ops.stitch_finalize(acc, cnt, end)


def replicate(model, n, group=1):
    """n - 1 more replicas of `model` (same config, same weights) for lib.dynamic_eval_many; with `group` = R > 1, n LOCKSTEP-GROUP
    models of R replicas each instead (`-kwargs lockstep=R`: R recordings advance through every window step in one batch per chain)."""
    if group > 1:
        if model.buffers:
            raise ValueError("lockstep groups: batch_renorm models keep the one-recording-per-chain path")
        out = []
        for _ in range(max(1, n)):
            m = SCConformerXL(dict(model.config), vocab_size=model.decoder.num_classes - 1, device=model.device, group=group)
            m.load_state_dict(model.state_dict())
            m.frozen = set(model.frozen)
            m.fused_attention = model.fused_attention
            m.eval()
            out.append(m)
        return out
    out = [model]
    for _ in range(max(0, n - 1)):
        m = SCConformerXL(dict(model.config), vocab_size=model.decoder.num_classes - 1, device=model.device)
        m.flat_params.copy_(model.flat_params)
        for name, buf in model.buffers.items():      # batch_renorm running statistics: part of the model, not of flat_params
            m.buffers[name].copy_(buf)
        m.frozen = set(model.frozen)
        m.fused_attention, m.fused_convmod = model.fused_attention, model.fused_convmod
        m.eval()
        out.append(m)
    return out


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    model, tokenizer = load_model_and_tokenizer(args, device)
    num_classes = model.decoder.num_classes
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=num_classes - 1, device=device)
    overlap = args.nsti_overlap
    min_frames = float(args.__dict__.get('min_minutes', 60.0)) * 60.0 * 100.0
    data_all = datasets_functions[args.dataset]("test") + datasets_functions[args.dataset]("dev")   # reference :59-63
    data = [el for el in data_all if el.get('frames', 0) >= min_frames]
    if rank == 0:
        print([el.get('frames', 0) / 6000.0 for el in data])
    chains = int(args.__dict__.get('chains', 1))
    lockstep = int(args.__dict__.get('lockstep', 1))     # -kwargs lockstep=R: R outer windows per chain advance in one batch (they are independent
    eval_fn = dynamic_eval if not args.awmc else AWMC    # recordings as far as eval_fn goes: weights restored, fresh optimiser, lib.py:494,636-637)
    if lockstep > 1 and not args.awmc:
        models = replicate(model, max(1, chains), group=lockstep)
    else:
        models = replicate(model, chains) if (chains > 1 and not args.awmc) else [model]
    mine = ddist.shard_longest_first([d.get('frames', 1) for d in data], world)[rank]

    wers = []
    for repeat in range(args.repeats):
        records = []
        for rec in mine:
            if rank == 0:
                print(f'Processing {rec + 1}/{len(data)}')
                print('\n-------\n' + data[rec]['id'] + '\n-------\n')
            audio_spec, gold_text = data[rec]['process_fn'](data[rec])
            spec_n = audio_spec.shape[-1]
            seq_len = args.nsti_seq_len if args.nsti_seq_len != -1 else spec_n
            ov = overlap
            if args.epochs == 0:            # just eval over the whole sequence if not performing NSTI
                seq_len, ov = spec_n, 0
            audio_dev = audio_spec.to(device=device, dtype=torch.float32)
            training_data, training_keys = prepare_chunks(audio_dev, seq_len, ov)
            keys = list(training_data.keys())
            chunks = [training_data[i] for i in keys]
            if len(models) > 1 or getattr(models[0], "R", 1) > 1:
                outs = lib.dynamic_eval_many(args, models, chunks, args.seq_len, args.overlap, tokenizer, use_tqdm=False,
                                             return_device=True)
            else:
                outs = [eval_fn(args, model, c, args.seq_len, args.overlap, tokenizer, beam_search_fn=None, use_tqdm=False,
                                return_device=True) for c in chunks]
            logits = outer_stitch([(k, o, c.shape[-1]) for k, o, c in zip(keys, outs, chunks)], ov, num_classes, device)
            out = normalize(decoder(logits)).lower()
            if rank == 0 and not args.not_verbose:
                print(gold_text, '\n', out, '\n\n')
            records.append({'index': rec, 'id': data[rec]['id'], 'hyp': out, 'gold': gold_text})
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
    parser.add_argument('-nsti_s', '--nsti_seq_len', type=int, default=-1, help='Sequence length for NSTI (-1 for full recording)')
    parser.add_argument('-nsti_o', '--nsti_overlap', type=int, default=0, help='Overlap for NSTI')
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
