"""ORACLE (test infrastructure only).  CPU restatement of the outer loop of BASELINE config 5, the reference's cross-dataset harness
(reference lcasr/run_cross_dataset_eval.py:92-218), statement by statement:
  baseline_args = args with epochs = 0                                  :91-94
  per repeat: baseline over every record of A, then of B (eval_fn with epochs 0 -> transcribe -> corpus WER)     :104-142
  for i in A: adapt on A[i] with return_params=True and the ADAPT overlap :147-156, load the updated parameters  :157-158,
              evaluate every record of B :160-176 and every record of A except i :178-195 (eval overlap, epochs 0),
              restore the original parameters :197-198
  results {a_baseline, b_baseline, a_to_b, a_to_a_loo}                    :200-209
`eval_fn` is called with the reference's own argument order, so the reference function extracted from lcasr/lib.py and
oracle.dynamic_eval_ref behind an adapter are interchangeable.  Pinned by tests/golden/loop_pins.json["cross"]: the reference's
statements :82-94 and :96-218 executed unchanged (tests/golden/make_loop_pins.py), with the un-vendored leaves (WER, normaliser, greedy decoder)
bound to the oracle's; what is pinned is therefore the outer loop's order, which overlap goes where, the leave-one-out index set and
the load / restore of the parameters.

One property of the reference's statements matters when they run on the CPU: `p.data = u.data.to(p.device)` (:157-158, :197-198) COPIES on a
CUDA device and ALIASES on the CPU (`.to('cpu')` of a CPU tensor returns the tensor itself).  After the restore of iteration i the live
parameters therefore share storage with `original_model_params`, the in-place optimiser updates of iteration i + 1 write through into that
list, and from then on "restore" reloads already-adapted weights.  `device_copies=False` reproduces that (it is what the pin was generated
with: the statements executed on the CPU), `device_copies=True` is the behaviour on a GPU — the reference's deployment and what the HIP
harness (dynamic-asr-eval_amd/run_cross_dataset_eval.py) implements.  Iteration i = 0 is identical in both."""
import argparse


def cross_dataset_ref(args, model, data_a, data_b, eval_fn, tokenizer, transcribe, word_error_rate_detail, record=None, device_copies=True):
    """`record` (optional list) receives (phase, i, hypotheses) for every scored corpus, in the order the reference scores them."""
    load = (lambda u: u.data.clone()) if device_copies else (lambda u: u.data)
    adapt_overlap = args.adapt_overlap if args.adapt_overlap is not None else args.overlap
    original_model_params = [p.clone().detach().cpu() for p in model.parameters()]
    args_dict = vars(args).copy()
    args_dict['epochs'] = 0
    baseline_args = argparse.Namespace(**args_dict)

    def score(records, idxs, phase, i):
        golds, preds = [], []
        for j in idxs:
            audio_spec, gold_text = records[j]['process_fn'](records[j])
            logits = eval_fn(baseline_args, model, audio_spec, args.seq_len, args.overlap, tokenizer, beam_search_fn=None)
            preds.append(transcribe(logits))
            golds.append(gold_text)
        if record is not None:
            record.append((phase, i, list(preds)))
        wer, words, ins_rate, del_rate, sub_rate = word_error_rate_detail(hypotheses=preds, references=golds)
        return {"wer": wer, "words": words, "ins_rate": ins_rate, "del_rate": del_rate, "sub_rate": sub_rate}

    out = []
    for repeat in range(args.repeats):
        a_to_b, a_to_a_loo = [], []
        a_baseline = score(data_a, range(len(data_a)), "a_baseline", None)
        b_baseline = score(data_b, range(len(data_b)), "b_baseline", None)
        for i in range(len(data_a)):
            audio_spec, _ = data_a[i]['process_fn'](data_a[i])
            _, updated_parameters = eval_fn(args, model, audio_spec, args.seq_len, adapt_overlap, tokenizer, beam_search_fn=None, return_params=True)
            for p, u in zip(model.parameters(), updated_parameters):
                p.data = load(u)
            a_to_b.append(score(data_b, range(len(data_b)), "a_to_b", i))
            a_to_a_loo.append(score(data_a, [k for k in range(len(data_a)) if k != i], "a_to_a_loo", i))
            for p, u in zip(model.parameters(), original_model_params):
                p.data = load(u)
        out.append({'a_baseline': a_baseline, 'b_baseline': b_baseline, 'a_to_b': a_to_b, 'a_to_a_loo': a_to_a_loo,
                    'repeat': f'{repeat + 1}/{args.repeats}'})
    return out


def oracle_eval_fn(optimizer_cls, lr_args_fn, specaug_fn, fixed_masks_fn):
    """Adapter: the reference's eval_fn signature (lcasr/lib.py:450-462) on oracle.dynamic_eval_ref.dynamic_eval_ref.
    `fixed_masks_fn(spec, seq_len, overlap)` -> {window key: masks}."""
    from .dynamic_eval_ref import dynamic_eval_ref

    def eval_fn(args, model, spec, seq_len, overlap, tokenizer, use_tqdm=False, beam_search_fn=None, return_params=False):
        assert beam_search_fn is None
        spec_n = spec.shape[-1]
        seq_len = seq_len if seq_len != -1 else args.config['audio_chunking']['size']
        if seq_len > spec_n:
            seq_len, overlap = spec_n, 0
        else:
            overlap = overlap if overlap != -1 else args.config['audio_chunking']['overlap']
        return dynamic_eval_ref(model, spec, seq_len, overlap, tokenizer, optimizer_cls, lr_args_fn(args), specaug_fn(args),
                                epochs=args.__dict__.get('epochs', 1), shuffle=args.__dict__.get('shuffle', False),
                                online=args.__dict__.get('online', False), fixed_masks=fixed_masks_fn(spec, seq_len, overlap),
                                return_params=return_params)
    return eval_fn
