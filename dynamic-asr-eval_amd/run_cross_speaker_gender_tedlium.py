"""Cross-speaker / cross-gender adaptation harness with the reference's flow, flags and pickle layout (reference
lcasr/run_cross_speaker_gender_tedlium.py:42-312): talks are selected and split into male / female groups by a speaker manifest
({'female': [{'talk_id': ...}], 'male': [...]}, :31-39, matched against `basename(rec['id'])` with the `.sph` suffix, :74-76);
baselines with epochs = 0 on both groups (:124-165); then for every male talk: adapt on it (return_params, :171-184), load the
adapted weights, score the other male talks and all female talks, restore (:185-226); the same with the groups swapped
(:228-286); results {male_baseline, female_baseline, male_to_male, male_to_female, female_to_female, female_to_male} (:288-297).

Here: adapted weights stay in HBM (one device-to-device copy of the flat buffer); the outer iterations are independent (weights
are restored after each), so ranks take them longest-first and the per-talk result lists are gathered to rank 0 in talk order;
baselines shard by talk with their edit counters all-reduced over RCCL.  `--dataset` defaults to the TEDLIUM-shaped synthetic set
(the corpus is not available offline)."""
import argparse
import json
import os
import pickle

import torch

from . import dist as ddist
from . import lib
from .datasets import datasets_functions
from .decoding import GreedyCTCDecoder
from .harness_common import clone_params, restore_params, score_texts, set_params, transcribe
from .run_dynamic_eval_full import load_model_and_tokenizer

DEFAULT_SPEAKER_MANIFEST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "speaker_manifest_15x15.json")


def load_speaker_manifest(path):
    """reference :31-39 -> (manifest, {talk_id + '.sph': 'F' | 'M'})"""
    with open(path, 'r') as f:
        manifest = json.load(f)
    speaker_gender = {}
    for row in manifest['female']:
        speaker_gender[row['talk_id'] + '.sph'] = 'F'
    for row in manifest['male']:
        speaker_gender[row['talk_id'] + '.sph'] = 'M'
    return manifest, speaker_gender


def _key(rec):
    base = os.path.basename(rec['id'])
    return base if base.endswith('.sph') else base + '.sph'


def main(args):
    assert args.split in ['test', 'dev', 'train'], f'Split must be either test, dev, or train (got {args.split})'
    rank, local_rank, world = ddist.init()
    device = torch.device('cuda', ddist.local_device_index(local_rank))
    torch.cuda.set_device(device)
    speaker_manifest, speaker_gender = load_speaker_manifest(args.speaker_manifest)
    if rank == 0:
        print(f"Loaded speaker manifest from {args.speaker_manifest}: {len(speaker_manifest['female'])} female, {len(speaker_manifest['male'])} male")
    model, tokenizer = load_model_and_tokenizer(args, device)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.decoder.num_classes - 1, device=device)
    fn = datasets_functions[args.dataset]
    all_data = fn('test') + fn('dev')                                  # reference :68-72 (test + dev + train)
    eval_data = [rec for rec in all_data if _key(rec) in speaker_gender]
    males = [rec for rec in eval_data if speaker_gender[_key(rec)] == 'M']
    females = [rec for rec in eval_data if speaker_gender[_key(rec)] == 'F']
    if rank == 0:
        print(f'Female data: {[os.path.basename(el["id"]) for el in females]}')
        print(f'Male data: {[os.path.basename(el["id"]) for el in males]}')
        print(f'Total data: {len(eval_data)}')
    assert len(females) + len(males) == len(eval_data), "Data filtered incorrectly"
    eval_fn = lib.dynamic_eval if not args.awmc else lib.AWMC
    adapt_overlap = args.adapt_overlap if args.adapt_overlap is not None else args.overlap
    if adapt_overlap != args.overlap and rank == 0:
        print(f'Using adapt_overlap={adapt_overlap} for adaptation (eval overlap={args.overlap})')
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

    def adapt_and_score(group, other):
        """for i in group: adapt on group[i], score group \\ {i} and all of other -> (group_to_group, group_to_other) in talk order"""
        shard = ddist.shard_longest_first([d.get('frames', 1) for d in group], world)[rank]
        mine = []
        for i in shard:
            audio_spec, _ = group[i]['process_fn'](group[i])
            _, updated = eval_fn(args, model, audio_spec, args.seq_len, adapt_overlap, tokenizer, beam_search_fn=None, use_tqdm=False,
                                 return_params=True, return_device=True)
            set_params(model, updated)
            same = score_texts(*evaluate(group, [k for k in range(len(group)) if k != i]))
            cross = score_texts(*evaluate(other, range(len(other))))
            mine.append({'index': i, 'same': same, 'cross': cross})
            restore_params(model, original)
        allr = ddist.gather_records(mine)
        return [r['same'] for r in allr], [r['cross'] for r in allr]

    for repeat in range(args.repeats):
        if rank == 0:
            print(f'\n=== Repeat {repeat + 1}/{args.repeats} ===')
            print('Male baseline')
        sm = ddist.shard_longest_first([d.get('frames', 1) for d in males], world)[rank]
        sf = ddist.shard_longest_first([d.get('frames', 1) for d in females], world)[rank]
        male_baseline = score_texts(*evaluate(males, sm), reduce_over_ranks=True)
        if rank == 0:
            print(f'Male baseline WER: {male_baseline["wer"]}')
            print('Female baseline')
        female_baseline = score_texts(*evaluate(females, sf), reduce_over_ranks=True)
        if rank == 0:
            print(f'Female baseline WER: {female_baseline["wer"]}')
            print('Male-X')
        male_to_male, male_to_female = adapt_and_score(males, females)
        if rank == 0:
            print('Female-X')
        female_to_female, female_to_male = adapt_and_score(females, males)
        if rank == 0:
            results = {'male_baseline': male_baseline, 'female_baseline': female_baseline, 'male_to_male': male_to_male,
                       'male_to_female': male_to_female, 'female_to_female': female_to_female, 'female_to_male': female_to_male,
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
    parser.add_argument('--dataset', '-d', type=str, default='synthetic_tedlium', choices=datasets_functions.keys())
    parser.add_argument('--repeats', '-r', type=int, default=1, help='Number of times to repeat the evaluation')
    parser.add_argument('--save_path', '-s', type=str, default='', help='path to save')
    parser.add_argument('--adapt_overlap', '-ao', type=int, default=None,
                        help='Overlap used during adaptation passes only. If unset, adaptation uses --overlap (current behavior).')
    parser.add_argument('--speaker_manifest', type=str, default=DEFAULT_SPEAKER_MANIFEST, help='Path to the gender selection manifest')
    return parser


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
