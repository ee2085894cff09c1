"""Teacher-prediction filters of the enc-dec adaptation loop: same flags (`add_enc_dec_teacher_filter_args`) and the same decision
function (`should_skip_faulty_teacher_prediction`) as the reference's lcasr/enc_dec_teacher_filters.py — pure host logic over the
teacher's token ids / text.  Decisions are pinned by tests/golden/reference_pins.json (the reference's own function executed via
ast extraction on a table of cases).  The sampled-decode agreement filter needs `word_error_rate_detail(use_cer=True)` of the
un-vendored `lcasr` package and sampled decoding: its flag is accepted and its text comparison is restated with this package's
edit distance on characters (parity unpinned); enc_dec_dynamic_eval refuses to run with it because sampling is out of scope."""
import re
from difflib import SequenceMatcher


def add_enc_dec_teacher_filter_args(parser):
    """reference lcasr/enc_dec_teacher_filters.py:7-125 (flag names, types and defaults)."""
    parser.add_argument('--teacher_filter_max_length', action='store_true')
    parser.add_argument('--teacher_min_frames_per_token', type=int, default=8)
    parser.add_argument('--teacher_filter_max_consecutive_token_repeat', action='store_true')
    parser.add_argument('--teacher_max_consecutive_token_repeat', type=int, default=3)
    parser.add_argument('--teacher_filter_repeated_token_ngrams', action='store_true')
    parser.add_argument('--teacher_repeated_token_ngram_sizes', type=int, nargs='+', default=[2, 3])
    parser.add_argument('--teacher_repeated_token_ngram_min_repeats', type=int, default=2)
    parser.add_argument('--teacher_filter_decode_agreement', action='store_true')
    parser.add_argument('--teacher_decode_agreement_temperature', type=float, default=0.7)
    parser.add_argument('--teacher_decode_agreement_min_similarity', type=float, default=0.65)
    parser.add_argument('--teacher_filter_low_confidence', action='store_true')
    parser.add_argument('--teacher_min_mean_max_prob', type=float, default=0.35)
    parser.add_argument('--teacher_max_mean_entropy', type=float, default=2.5)
    parser.add_argument('--teacher_filter_repeated_words', action='store_true')
    parser.add_argument('--teacher_max_consecutive_word_repeat', type=int, default=3)
    parser.add_argument('--teacher_filter_ctc_agreement', action='store_true')
    parser.add_argument('--teacher_ctc_agreement_min_similarity', type=float, default=0.5)
    return parser


def _sequence_similarity(first, second):
    return SequenceMatcher(a=list(first), b=list(second)).ratio()


def _text_cer_similarity(hyp_text, ref_text):
    if not hyp_text and not ref_text:
        return 1.0
    if not hyp_text or not ref_text:
        return 0.0
    from .wer import _align
    vocab = {}
    h = [vocab.setdefault(c, len(vocab)) for c in hyp_text]
    r = [vocab.setdefault(c, len(vocab)) for c in ref_text]
    i, d, s = _align(h, r)
    return max(0.0, 1.0 - (i + d + s) / len(r))


def _word_sequence(text):
    return re.findall(r"[a-z0-9']+", text.lower())


def _longest_consecutive_repeat(sequence):
    longest, longest_item, current, previous = 0, None, 0, None
    for item in sequence:
        if item == previous:
            current += 1
        else:
            previous, current = item, 1
        if current > longest:
            longest, longest_item = current, item
    return longest, longest_item


def _find_repeated_ngram_loop(sequence, ngram_size, min_repeats):
    span = ngram_size * min_repeats
    if ngram_size <= 0 or min_repeats <= 1 or len(sequence) < span:
        return False, (), 0
    for start in range(len(sequence) - span + 1):
        ngram = tuple(sequence[start:start + ngram_size])
        count, cursor = 1, start + ngram_size
        while cursor + ngram_size <= len(sequence) and tuple(sequence[cursor:cursor + ngram_size]) == ngram:
            count += 1
            cursor += ngram_size
        if count >= min_repeats:
            return True, ngram, count
    return False, (), 0


def should_skip_faulty_teacher_prediction(args, teacher_pred_tokens, teacher_pred_text, spec_frames, agreement_text=None,
                                          teacher_mean_max_prob=None, teacher_mean_entropy=None, ctc_text=None):
    """reference lcasr/enc_dec_teacher_filters.py `should_skip_faulty_teacher_prediction`: -> (skip, reason), filters tried in the
    reference's order: max length, consecutive token repeat, token n-gram loops, decode agreement, low confidence, repeated words,
    CTC agreement."""
    g = args.__dict__.get
    if g('teacher_filter_max_length', False):
        min_fpt = g('teacher_min_frames_per_token', 8)
        if min_fpt > 0:
            max_tokens = spec_frames / min_fpt
            if len(teacher_pred_tokens) > max_tokens:
                return True, (f'too many teacher tokens ({len(teacher_pred_tokens)} tokens for {spec_frames} frames; '
                              f'max {max_tokens:.2f})')
    if g('teacher_filter_max_consecutive_token_repeat', False):
        longest, token = _longest_consecutive_repeat(teacher_pred_tokens)
        limit = g('teacher_max_consecutive_token_repeat', 3)
        if longest > limit:
            return True, f'teacher token {token} repeated {longest} times consecutively (limit {limit})'
    if g('teacher_filter_repeated_token_ngrams', False):
        min_repeats = g('teacher_repeated_token_ngram_min_repeats', 2)
        for n in sorted(set(g('teacher_repeated_token_ngram_sizes', [2, 3]))):
            repeated, ngram, count = _find_repeated_ngram_loop(teacher_pred_tokens, n, min_repeats)
            if repeated:
                return True, f'teacher token {n}-gram {list(ngram)} repeated {count} times consecutively'
    if g('teacher_filter_decode_agreement', False) and agreement_text is not None:
        min_sim = g('teacher_decode_agreement_min_similarity', 0.65)
        sim = _text_cer_similarity(agreement_text, teacher_pred_text)
        if sim < min_sim:
            return True, f'teacher decode agreement too low (1-CER={sim:.2f} < {min_sim:.2f})'
    if g('teacher_filter_low_confidence', False):
        min_p, max_h = g('teacher_min_mean_max_prob', 0.35), g('teacher_max_mean_entropy', 2.5)
        if teacher_mean_max_prob is not None and teacher_mean_max_prob < min_p:
            return True, f'teacher mean max prob too low ({teacher_mean_max_prob:.3f} < {min_p:.3f})'
        if teacher_mean_entropy is not None and teacher_mean_entropy > max_h:
            return True, f'teacher mean entropy too high ({teacher_mean_entropy:.3f} > {max_h:.3f})'
    if g('teacher_filter_repeated_words', False):
        longest, word = _longest_consecutive_repeat(_word_sequence(teacher_pred_text))
        limit = g('teacher_max_consecutive_word_repeat', 3)
        if longest > limit:
            return True, f'teacher word "{word}" repeated {longest} times consecutively (limit {limit})'
    if g('teacher_filter_ctc_agreement', False) and ctc_text is not None:
        min_sim = g('teacher_ctc_agreement_min_similarity', 0.5)
        sim = _sequence_similarity(_word_sequence(teacher_pred_text), _word_sequence(ctc_text))
        if sim < min_sim:
            return True, f'encoder-decoder/ctc agreement too low ({sim:.2f} < {min_sim:.2f}); ctc="{ctc_text}"'
    return False, ''
