#!/usr/bin/env python3
"""bench.py — dynamic-eval throughput on MI355X (metric and workload of BASELINE.json).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: starts its N ranks itself, self_launch())
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP = one pass of the hot path over one synthetic recording of the Earnings-22 long-form shape
(BASELINE.json configs[1]: 1 h = 360 000 log-mel frames, seq_len 16384 / overlap 14336 -> 169 windows): for every window
forward (B=2: augmented + clean copy) -> on-device greedy pseudo-label -> CTC loss + gradient -> backward -> one MADGRAD
step; then the final no-grad pass over all windows, the on-device stitch and the final greedy decode
(reference lcasr/lib.py:450-640 driven by run_dynamic_eval_full.py:84-100).  The recording is resident in HBM when the
timed region starts.  Weak scaling: every rank adapts on its own recordings (they are independent: weights restored,
fresh optimiser per call); the only collective is the WER-counter all-reduce after the timed region.

Prints ONE JSON line on rank 0 with `roofline` (fp32-MFMA GEMM family, measured live with HIP events on the launch
stream) and, at N=1, `cpu_baseline` (the CPU oracle restatement timed on the host cores on a bounded sample: ONE weight-carrying
recording of 8 windows + tail), `parity` (that recording through the HIP path with the same weights and masks, offline and online,
compared with the oracle's outputs), `other_workloads` (AWMC, wav2vec2 dynamic_eval_su, enc-dec teacher_ce: audio-s/s of the path's other loops),
`value_degenerate_labels` (the model's own collapsing pseudo-labels instead of --label_tokens seeded ids per window),
`value_boundary` (recordings start in host memory, log-probs come back as numpy: the reference's call contract) and `value_online`
(`online=True`: the adapt loop's own clean-copy posteriors are stitched, no final pass — the mode of the reference's published timing)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 256 CUs x 2.4 GHz


def progress(msg):
    """Stage marker on STDERR (stdout carries exactly one JSON line): a 20-step run is silent for 7 - 8 minutes otherwise, which a watchdog on
    the GPU box takes for a hang."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)      # two lockstep groups of 4: one per chain, no partial group
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=3600.0, help="length of each synthetic recording")
    ap.add_argument("--seq_len", type=int, default=16384)
    ap.add_argument("--overlap", type=int, default=14336)
    ap.add_argument("--vocab", type=int, default=4095)
    ap.add_argument("--lr", type=float, default=9e-5)
    ap.add_argument("--online", type=int, default=0)
    ap.add_argument("--blank_bias", type=float, default=-1.0, help="<0: calibrate for a speech-like token rate")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_windows", type=int, default=8, help="full windows of the ONE weight-carrying recording the CPU oracle runs "
                    "(16384 / 14336: cpu_windows full windows + the short tail window); the first adapt step is the untimed warm-up")
    ap.add_argument("--side_workloads", type=int, default=1, help="1: after the timed region (never part of `value`) also time the other "
                    "paths of the same C-ABI on short synthetic inputs and print their audio-s/s under `other_workloads`: AWMC "
                    "(lcasr/lib.py:206-376), wav2vec2-base dynamic_eval_su (wav2vec2/lib.py:293-462), enc-dec teacher_ce (lcasr/lib.py:1475-1732)")
    ap.add_argument("--rendezvous_only", action="store_true", help="start the ranks, run the bench's barrier + two collectives, print "
                    "one JSON line and exit without touching the GPU (launch-path test: tests/test_host_cpu.py)")
    ap.add_argument("--label_tokens", type=int, default=400, help="pseudo-label ids per window in the timed region: a seeded model self-training "
                    "on noise collapses to the empty transcript within a few windows (degenerate CTC lattice, L = 1), so tokenizer.encode "
                    "returns a fixed seeded sequence of this many ids per window (speech-like: ~2.4 tokens/s) and the lattice runs at "
                    "L = 2 * N + 1; 0 = the model's own (collapsing) pseudo-labels, reported separately as value_degenerate_labels")
    ap.add_argument("--side_steps", type=int, default=-1, help="recordings per side measurement (value_degenerate_labels, value_boundary); "
                    "-1 = one round of the chains, 0 = skip them")
    ap.add_argument("--graphs", type=int, default=1, help="hipGraph replay of the per-window launch sequences")
    ap.add_argument("--prewarm_s", type=float, default=40.0, help="seconds of the untimed workload BEFORE the W warm-up steps: the first "
                    "process on an idle MI355X reads 3-10 %% low until the GPU has been under this load for some tens of seconds "
                    "(743 -> 772 audio-s/s with 40 s; 12 s of plain GEMMs do not do it); steady state is what is reported")
    ap.add_argument("--sample_every", type=int, default=32, help="every n-th window step runs eagerly for the live roofline sampling "
                    "(costs 0.7 %% of the throughput at 32: 771 vs 776 audio-s/s at 96, 775 with sampling off)")
    ap.add_argument("--final_batch", type=int, default=4, help="windows per forward in the final pass (they are independent)")
    ap.add_argument("--pcie", type=int, default=0, help="1: recordings start in HOST memory and results come back as numpy (the "
                    "PCIe-inclusive rate quoted in DESIGN.md; never the headline value)")
    ap.add_argument("--chains", type=int, default=2, help="independent chains in flight per GPU (own stream + model object each)")
    ap.add_argument("--lockstep", type=int, default=4, help="R > 1: every chain is a LOCKSTEP GROUP of R recordings advancing through the same window "
                    "step in one batch (SCConformerXL(group=R): one launch per layer for all R, each recording with its own weights); chains x R "
                    "recordings in flight.  Default 2 chains x 4 (r04, driver form on one box: 805 audio-s/s against 791 for 3 chains x 1, "
                    "profiles/r04_ab_lockstep_driver_form.log); --lockstep 1 --chains 3 is the r03 configuration")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment (the driver's own command form): start the N ranks as
    children — one process per GPU, `python -m torch.distributed.run` on 127.0.0.1 — BEFORE this process has made any GPU call (a
    process that has initialised HIP must never be replaced or forked into ranks), relay their output (the children inherit
    stdout / stderr, so rank 0's JSON line is this command's JSON line) and return their exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(a):
    """The N-rank launch path without the workload: process group, barrier, the max-over-ranks reduction of the timing and the
    counter all-reduce that bench.py runs around / after its timed region."""
    from dynamic_asr_eval_amd import dist as ddist
    rank, local_rank, world = ddist.init()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    ddist.barrier()
    slowest = ddist.max_over_ranks(float(rank))
    seen = ddist.all_reduce_counts((1, rank, 0, 0))
    ddist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous": True, "n_gpus": world, "ranks_seen": int(seen[0]), "rank_sum": int(seen[1]), "max_rank": slowest}), flush=True)
    ddist.shutdown()


def pmc_traffic():
    """HBM bytes per gemm_f32_kernel launch from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes of this same command; scripts/pmc_summary.py).  bench.py cannot run the profiler on itself, so this is
    the latest committed measurement, or null when none exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json")))
    if not files:
        return None
    try:
        return round(json.load(open(files[-1]))["hbm_bytes_per_launch"])
    except Exception:
        return None


def pick_chains(max_chains, steps):
    """Chains per GPU for `steps` recordings: with c chains the recordings run in rounds of c; a last round with a single
    recording runs at one-chain speed (measured: 2 or more in flight cost ~0.855 of a lone recording each), so prefer the chain
    count whose last round is not a lone recording (e.g. 4 steps -> 2 chains rather than 3 + 1)."""
    best, best_cost = 1, float("inf")
    for c in range(1, max(1, min(max_chains, steps)) + 1):
        full, rem = divmod(steps, c)
        per = lambda k: k * (1.0 if k == 1 else 0.855)
        cost = full * per(c) + (per(rem) if rem else 0.0)
        if cost <= best_cost + 1e-9:
            best, best_cost = c, cost
    return best


def make_args(a):
    ns = argparse.Namespace()
    ns.config = {'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {'max_seq_len': 0}}
    ns.__dict__.update(dict(optim_lr=a.lr, epochs=1, shuffle=False, online=bool(a.online), quiet=True,
                            spec_augment_n_freq_masks=6, spec_augment_freq_mask_param=34, spec_augment_n_time_masks=0,
                            use_graphs=bool(a.graphs), final_pass_batch=a.final_batch))
    return ns


class SubstituteLabelTokenizer:
    """SyntheticTokenizer whose encode() returns a FIXED seeded id sequence of `n` tokens (bench only): the decode -> text -> encode
    hop, the pinned upload and the CTC lattice then carry the sizes a real checkpoint produces (reference lcasr/lib.py:565-575)."""

    def __init__(self, base, n, seed=4321):
        self.base, self.n = base, int(n)
        g = torch.Generator().manual_seed(seed)
        self.ids = torch.randint(0, base.vocab_size(), (self.n,), generator=g).tolist()
        self.text = base.decode(self.ids)

    def vocab_size(self):
        return self.base.vocab_size()

    def decode(self, ids):
        return self.base.decode(ids)

    def encode(self, text):
        self.base.encode(text)                      # the model's own pseudo-label text still makes the round trip
        return self.base.encode(self.text)


def cpu_baseline(a, hip_model, dev):
    """CPU oracle (oracle/dynamic_eval_ref.py + oracle/conformer_ref.py) on a bounded sample: ONE recording of `cpu_windows` full
    16384-frame windows + the short tail window at the benchmark's own seq_len / overlap, so the weights and the MADGRAD state carry
    from window to window exactly as in the 1 h job.  Per window the oracle does an adapt step (B=2 fwd + CTC + bwd + MADGRAD) and a
    final-pass forward + stitch; the 1 h recording is 169 such windows, so audio-s/s = 3600 / (169 * (adapt + final) seconds per
    window), the first adapt step left out as the warm-up.
    The SAME recording then goes through the HIP path (lib.dynamic_eval, same weights, same stored SpecAugment masks, the model's
    own pseudo-labels), offline and online: `parity` = the largest |log-prob difference| of the adapted, stitched outputs, the
    per-band curve of the online run (band k = rows stitched from the clean posteriors after ~k adapt steps), argmax agreement."""
    import numpy as np
    from oracle.conformer_ref import SCConformerXLRef
    from oracle.dynamic_eval_ref import draw_masks, dynamic_eval_ref, greedy_ctc_ids, prepare_chunks
    from oracle.madgrad_ref import MADGRAD
    from dynamic_asr_eval_amd import lib
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    from dynamic_asr_eval_amd.wer import edit_counts
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("DYN_CPU_BASELINE_THREADS", 16)))  # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    model = SCConformerXLRef(vocab_size=a.vocab, seed=0, blank_bias=0.0)
    model.load_state_dict({k: v.cpu() for k, v in hip_model.state_dict().items()})    # exactly the weights the HIP path runs
    tok = SyntheticTokenizer(a.vocab)
    stride = a.seq_len - a.overlap
    spec = synthetic_spec(a.seq_len + (a.cpu_windows - 1) * stride, seed=1234)
    _, keys = prepare_chunks(spec, a.seq_len, a.overlap)
    mg = torch.Generator().manual_seed(99)
    masks = {k: (draw_masks(6, 34, 80, mg), ([], [])) for k in keys}
    tm = {}
    want, want_online = dynamic_eval_ref(model, spec, a.seq_len, a.overlap, tok, MADGRAD, {'lr': a.lr}, {}, epochs=1, online=False,
                                         fixed_masks=masks, timings=tm, also_online=True)
    adapt, final = tm['adapt'][1:], tm['final'][1:]            # the first window of each loop is the warm-up
    med = lambda v: sorted(v)[len(v) // 2]
    dt = med(adapt) + med(final)                               # median per window: the second window still runs warm-up-ish
    n_windows = 169
    per = [x + y for x, y in zip(adapt, final)]
    base = {"value": round(3600.0 / (n_windows * dt), 4), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "seconds_per_window": [round(t, 2) for t in per], "spread": round((max(per) - min(per)) / dt, 3),
            "sample": f"{len(adapt)} of {n_windows} windows of ONE weight-carrying recording ({len(keys)} windows, seq_len {a.seq_len} / overlap "
                      f"{a.overlap}; the first window is the warm-up): MEDIAN per window of B=2 forward + CTC + backward + MADGRAD ({med(adapt):.1f} s) "
                      f"and of the final-pass forward + stitch ({med(final):.1f} s) on {cores} host threads, extrapolated x{n_windows}"}
    # the same recording through the HIP path
    def run(online, epochs=1):
        args = make_args(a)
        args.spec_augment_fixed_masks, args.online, args.epochs = masks, online, epochs
        R = getattr(hip_model, "R", 1)
        if R > 1:      # the timed path: a lockstep group — the same recording on every replica.  The replicas are the same arithmetic up to the GEMM
            # planner's K-slicing of the LAST partial round of tiles (those tiles belong to the last batch entries: another summation order),
            # so they agree to rounding, not bit for bit; the spread between replicas is reported next to the distance from the oracle
            outs = lib.dynamic_eval_lockstep(args, hip_model, [spec.to(dev)] * R, a.seq_len, a.overlap, tok, use_tqdm=False)
            if epochs:
                spread.append(max(float(np.abs(outs[0] - o).max()) for o in outs[1:]))
            return outs[0]
        return lib.dynamic_eval(args, hip_model, spec.to(dev), a.seq_len, a.overlap, tok, use_tqdm=False)
    spread = []
    got, got_online = run(False), run(True)
    band = stride // 8

    def cmp(g, w):
        d = np.abs(g - w).max(-1)
        bad = g.argmax(-1) != w.argmax(-1)
        margin = None
        if bad.any():
            top2 = np.sort(w[bad], -1)[:, -2:]
            margin = float((top2[:, 1] - top2[:, 0]).max())
        return d, int(bad.sum()), margin
    d_off, bad_off, m_off = cmp(got, want)
    d_on, bad_on, m_on = cmp(got_online, want_online)
    hyp_h, hyp_o = tok.decode(greedy_ctc_ids(torch.from_numpy(got), a.vocab)), tok.decode(greedy_ctc_ids(torch.from_numpy(want), a.vocab))
    # the adapted transcripts of a seeded model are (nearly) empty, so the decoded-text comparison is also made BEFORE any adaptation
    # (epochs = 0), where the calibrated model emits a few hundred tokens per window
    want0 = dynamic_eval_ref(model, spec, a.seq_len, a.overlap, tok, MADGRAD, {'lr': a.lr}, {}, epochs=0, online=False)
    got0 = run(False, epochs=0)
    h0, o0 = greedy_ctc_ids(torch.from_numpy(got0), a.vocab), greedy_ctc_ids(torch.from_numpy(want0), a.vocab)
    c0 = list(edit_counts([tok.decode(h0)], [tok.decode(o0)]))
    unadapted = {"max_abs_dlogp": float(f"{float(np.abs(got0 - want0).max()):.3e}"), "argmax_equal": bool((got0.argmax(-1) == want0.argmax(-1)).all()),
                 "words": len(o0), "wer_counters_hip_vs_oracle": c0}
    ca = list(edit_counts([hyp_h], [hyp_o]))
    parity = {"max_abs_dlogp": float(f"{float(d_off.max()):.3e}"), "max_abs_dlogp_online": float(f"{float(d_on.max()):.3e}"),
              "argmax_equal": bad_off == 0 and bad_on == 0, "argmax_mismatch_frames": [bad_off, bad_on], "frames": [int(d_off.shape[0]), int(d_on.shape[0])],
              "largest_oracle_margin_at_a_mismatch": max([m for m in (m_off, m_on) if m is not None], default=None),
              "windows": len(keys), "adapt_steps_carried": len(keys),
              "replica_spread_offline_online": [float(f"{v:.3e}") for v in spread] if spread else None,
              "online_max_abs_dlogp_per_band": [float(f"{float(d_on[k:k + band].max()):.2e}") for k in range(0, d_on.shape[0], band)],
              "band_rows": band, "words": len(hyp_o.split()), "wer_counters_hip_vs_oracle": ca, "unadapted": unadapted,
              "what": "adapted + stitched log-probs of ONE weight-carrying recording (the cpu_baseline sample): HIP path vs CPU oracle, same weights and "
                      "SpecAugment masks, offline (final pass) and online (band k of the online curve = rows stitched after ~k adapt steps); "
                      "wer_counters = (ins, del, sub, words) of the HIP transcript against the oracle's over `words` words (a seeded model adapts "
                      "towards the empty transcript, so the content-bearing transcript comparison is `unadapted`).  Reading the numbers: the CTC lattice is "
                      "bit-identical to torch's CPU kernel on identical inputs (tests/test_ops_gpu.py::test_ctc_lattice_is_bitwise_torch_cpu), so what separates the two "
                      "runs is the ~1e-6 GEMM-order difference of their log-probs going through a chaotic fp32 recursion: the first adapt step trains on a ~400-token "
                      "label on noise (|alpha| ~ 3000 - 4600, ulp 2.4e-4 - 4.9e-4) and leaves a 1e-3 - 1e-2 peak at online band 9, after which the labels are empty and "
                      "the difference decays; the fp32 oracle is as far from its own float64 run (DESIGN.md section 4, profiles/r04_drift_*.json)"}
    return base, parity


class _Skip(Exception):
    pass


def cpu_baseline_soft_dtw(x, y, gamma, value, B):
    """CPU leg of the soft-DTW figure (part of the cpu_baseline leg: the only place bench.py may touch oracle/): the numpy fp64 restatement
    of the reference's Numba recurrences timed on a bounded sample (<= 4 pairs of the batch, scaled to B), and the HIP value checked against it."""
    import numpy as np
    from oracle import softdtw_ref
    nb = x.shape[0]
    t = time.perf_counter()
    D = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)
    R = softdtw_ref.softdtw_forward(D, gamma)
    softdtw_ref.softdtw_backward(D, R, gamma)
    cpu = (time.perf_counter() - t) * B / nb
    return cpu, bool(np.allclose(R[:, -2, -2], value[:nb], rtol=1e-5, atol=1e-5))


def soft_dtw_latency(dev):
    """Soft-DTW (reference wav2vec2/soft_dtw_cuda.py; constructed wav2vec2/lib.py:130,370 as SoftDTW(use_cuda=True, gamma=1.5)): forward +
    backward latency through the module, measured the way the reference's own `profile()` does (:383-418: forward, then autograd.grad w.r.t.
    the first argument; first iteration dropped) at the config-3 shape [2, 409, 32] (131 072-sample windows -> 409 frames of 32 logits) and at
    the reference's three self-check shapes (:426-428), next to the CPU oracle (numpy fp64 restatement of the Numba recurrences) on a bounded
    sample of the batch.  The scan is latency-bound (N + M - 1 anti-diagonals, one barrier each): `us_per_diagonal` is the figure that
    matters, `hbm_frac` (algorithmic bytes: D read twice, the fp64 lattice written once and read once, E written) says how far from
    bandwidth it is."""
    import numpy as np
    from dynamic_asr_eval_amd.soft_dtw import SoftDTW
    rows = []
    for B, N, M, d, gamma in ((2, 409, 409, 32, 1.5), (128, 17, 15, 2, 1.0), (512, 64, 64, 2, 1.0), (512, 256, 256, 2, 1.0)):
        sd = SoftDTW(True, gamma=gamma, normalize=False)
        g = torch.Generator().manual_seed(1234)
        ts = []
        for i in range(6):
            x = torch.rand(B, N, d, generator=g).to(dev).requires_grad_()
            y = torch.rand(B, M, d, generator=g).to(dev)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            val = sd(x, y)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            grad = torch.autograd.grad(val, x, grad_outputs=torch.ones_like(val))[0]
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            if i:
                ts.append((t1 - t, t2 - t1))
        fwd, bwd = float(np.mean([v[0] for v in ts])), float(np.mean([v[1] for v in ts]))
        nb = min(B, 4)                                                   # bounded CPU sample: <= 4 pairs of the batch, scaled to B
        cpu, ok = cpu_baseline_soft_dtw(x.detach().cpu().double().numpy()[:nb], y.cpu().double().numpy()[:nb], gamma, val.detach().cpu().numpy(), B)
        alg = B * (2 * N * M * 4 + 2 * (N + 2) * (M + 2) * 8 + N * M * 4)
        rows.append({"shape": [B, N, M, d], "gamma": gamma, "fwd_ms": round(fwd * 1e3, 3), "bwd_ms": round(bwd * 1e3, 3),
                     "us_per_diagonal": round((fwd + bwd) * 1e6 / (2 * (N + M - 1)), 3), "algorithmic_bytes": alg,
                     "hbm_frac": round(alg / (fwd + bwd) / 8e12, 5), "cpu_oracle_ms": round(cpu * 1e3, 1), "cpu_sample": f"{nb} of {B} pairs, scaled",
                     "value_matches_oracle": ok})
    return rows


def other_workloads(a, model, dev, which=('awmc', 'wav2vec2_su', 'enc_dec_teacher_ce', 'soft_dtw')):
    """Driver-visible throughput of the path's other loops (one recording each, one chain, second of two runs; untimed as far as
    `value` goes).  Same kernels, same C-ABI; shapes: AWMC on a 10-min recording of the benchmark's model and window; wav2vec2-base
    (Wav2Vec2Config() defaults, seeded) `dynamic_eval_su` over a 5-min TEDLIUM-shape talk cut by the reference's fetch_utterances
    rule; enc-dec `teacher_ce` (6 x 768 encoder + 2 x 256 decoder, seeded) on a 5-min recording, 2048-frame windows."""
    import io
    from contextlib import redirect_stdout
    from dynamic_asr_eval_amd import lib
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    out = {}

    def best_of_two(fn):
        ts = []
        for _ in range(2):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            with redirect_stdout(io.StringIO()):       # the mirrored loops print what the reference prints
                fn()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t)
        return min(ts)

    try:
        if 'awmc' not in which:
            raise _Skip
        secs = 600.0
        spec = synthetic_spec(int(secs * 100), seed=4242).to(dev)
        args = make_args(a)
        tok = SyntheticTokenizer(a.vocab)
        dt = best_of_two(lambda: lib.AWMC(args, model, spec, a.seq_len, a.overlap, tok, use_tqdm=False, return_device=True))
        out["awmc"] = {"value": round(secs / dt, 1), "unit": "audio-s/s", "sample": f"{secs:.0f} s recording, seq_len {a.seq_len} / overlap {a.overlap}, 1 chain"}
    except _Skip:
        pass
    except Exception as e:                              # a side measurement must never take the headline down with it
        out["awmc"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:
        if 'wav2vec2_su' not in which:
            raise _Skip
        from dynamic_asr_eval_amd import run_wav2vec2 as RW, wav2vec2_lib as W
        from dynamic_asr_eval_amd.wav2vec2_model import Wav2Vec2ForCTC
        wm = Wav2Vec2ForCTC(None, device=dev)
        RW.init_synthetic(wm, 0)
        wm.eval()
        talks = [RW.fetch_utterances_synthetic(300.0, 7 + 17 * t) for t in range(4)]        # the reference's driver walks the talks of the split
        utts = talks[0]
        audio_s = sum(u['waveform'].shape[-1] for u in utts) / 16000.0
        audio_all = sum(u['waveform'].shape[-1] for t in talks for u in t) / 16000.0
        tokw = W.CharTokenizer()

        def su(m=wm, **kw):
            return W.dynamic_eval_su(argparse.Namespace(epochs=1, shuffle=False, **kw), m, [dict(u) for u in utts], 0, 0, tokw, None, use_tqdm=False,
                                     optim=W.MADGRAD, lr_args={'lr': 1e-6})
        dt_eager = best_of_two(lambda: su(use_graphs=False))            # every utterance launched kernel by kernel at its own length (r01 - r03)
        with redirect_stdout(io.StringIO()):
            su(); su()                                                   # a length bucket is captured the second time it is seen
        dt_one = best_of_two(su)                                        # one talk at a time, bucket graphs replaying
        n_buckets, held = len(wm._graphs), wm.graph_bytes()
        # talks are independent (weights restored per call, wav2vec2/lib.py:455-460): two in flight, one model replica + stream each
        n_chains = int(os.environ.get("DYN_W2V_CHAINS", "2"))
        wm.graph_after = 1
        chain_models = RW.replicate(wm, n_chains)
        margs = argparse.Namespace(epochs=1, shuffle=False)

        def many():
            return W.dynamic_eval_su_many(margs, chain_models, [[dict(u) for u in t] for t in talks], 0, 0, tokw, None, optim=W.MADGRAD, lr_args={'lr': 1e-6})
        with redirect_stdout(io.StringIO()):
            for m in chain_models:                                       # every replica has met (= captured) every bucket before the timed passes
                for t in talks:
                    W.dynamic_eval_su(margs, m, [dict(u) for u in t], 0, 0, tokw, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-6})
        dt = best_of_two(many)
        # roofline entry of config 3's loop: every matrix product of one more (untimed, eager) pass counted, over the timed pass's wall time
        from dynamic_asr_eval_amd import ops as _ops
        _ops.gemm_profile_start(every=1 << 30)
        with redirect_stdout(io.StringIO()):
            su(use_graphs=False)
        torch.cuda.synchronize(dev)
        pr = _ops.gemm_profile_stop()
        tf = (pr["flops"] + pr["attn_flops"]) * (audio_all / audio_s) / dt / 1e12
        out["wav2vec2_su"] = {"value": round(audio_all / dt, 1), "unit": "audio-s/s", "value_one_talk": round(audio_s / dt_one, 1),
                              "value_eager": round(audio_s / dt_eager, 1),
                              "sample": f"wav2vec2-base shape; `value`: {len(talks)} talks ({sum(len(t) for t in talks)} utterances, {audio_all:.0f} s of 16 kHz audio), {n_chains} in "
                                        f"flight (one replica + stream each), hipGraph replay over length buckets of {wm.bucket_frames} frames, all captured before "
                                        f"the timed pass; `value_one_talk`: one talk ({len(utts)} utterances, {audio_s:.0f} s) at a time, {n_buckets} buckets "
                                        f"({held / 2 ** 30:.1f} GiB held); `value_eager`: the same launched kernel by kernel at every utterance's own length",
                              "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "gemm_launches_per_utterance": round(pr["calls"] / len(utts), 1),
                                           "gflop_per_utterance": round(pr["flops"] / len(utts) / 1e9, 1),
                                           "note": "whole-loop figure (every kernel and host gap of the pass in the denominator; flops of the unpadded utterances, "
                                                   "scaled from the first talk by audio length): the loop is GPU-bound (kernel time = wall time, "
                                                   "profiles/r04_kernel_stats_wav2vec2_su.csv) by products of 40 - 500 tiles each, latency- not throughput-bound"}}
        del chain_models
        del wm
    except _Skip:
        pass
    except Exception as e:
        out["wav2vec2_su"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:
        if 'enc_dec_teacher_ce' not in which:
            raise _Skip
        from dynamic_asr_eval_amd.enc_dec import EncDecSCConformerXL, enc_dec_dynamic_eval
        from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
        em = EncDecSCConformerXL({}, vocab_size=a.vocab, device=dev)
        init_synthetic(em, seed=1, blank_bias=1.0)
        secs = 300.0
        spec = synthetic_spec(int(secs * 100), seed=4243).to(dev)
        eargs = make_args(a)
        eargs.training_mode = 'teacher_ce'
        dt = best_of_two(lambda: enc_dec_dynamic_eval(eargs, em, spec, 2048, 0, SyntheticTokenizer(a.vocab), use_tqdm=False))
        out["enc_dec_teacher_ce"] = {"value": round(secs / dt, 1), "unit": "audio-s/s", "sample": f"{secs:.0f} s recording, 2048-frame windows, KV-cached greedy teacher + final decode"}
        del em
    except _Skip:
        pass
    except Exception as e:
        out["enc_dec_teacher_ce"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:
        if 'soft_dtw' in which:
            out["soft_dtw"] = soft_dtw_latency(dev)
    except Exception as e:
        out["soft_dtw"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:       # not under torchrun: become the launcher (nothing has touched the GPU yet)
        sys.exit(self_launch(a))
    if a.rendezvous_only:
        return rendezvous_only(a)
    from dynamic_asr_eval_amd import dist as ddist, lib, ops
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    from dynamic_asr_eval_amd.decoding import GreedyCTCDecoder
    from dynamic_asr_eval_amd.model import SCConformerXL
    from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer

    rank, local_rank, world = ddist.init()
    assert world == a.gpus or world == 1 and a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", ddist.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    R = max(1, a.lockstep)
    n_chains = pick_chains(a.chains, (a.steps + R - 1) // R)
    models = []
    for _ in range(n_chains):
        m = SCConformerXL(vocab_size=a.vocab, device=dev, group=R)
        init_synthetic(m, seed=0, blank_bias=0.0)
        if R > 1:
            m.load_state_dict(m.state_dict())          # replica 0's seeded weights into every replica of the group
        models.append(m)
    model = models[0]
    plain_tok = SyntheticTokenizer(a.vocab)
    tok = SubstituteLabelTokenizer(plain_tok, a.label_tokens) if a.label_tokens > 0 else plain_tok
    args = make_args(a)
    n_frames = int(a.seconds * 100)
    torch.manual_seed(1234 + rank)

    def one_step(step_idx):
        spec = synthetic_spec(n_frames, seed=1234 + 1000 * rank + step_idx)
        return spec.pin_memory() if a.pcie else spec.to(dev)          # default: resident in HBM before timing

    decoder = GreedyCTCDecoder(tok, blank_id=a.vocab, device=dev)

    def run_many(spec_list, tokenizer=None, pcie=None, online=None, group_sizes=None):
        """`n_chains` recordings in flight: one stream + one model replica each, advanced round-robin by one host thread."""
        pcie = a.pcie if pcie is None else pcie
        run_args = args
        if online is not None and bool(online) != bool(a.online):
            run_args = argparse.Namespace(**vars(args)); run_args.online = bool(online)
        n_groups = len(group_sizes) if group_sizes and R > 1 else len(lib.lockstep_group_sizes(len(spec_list), R, len(models)))
        outs = lib.dynamic_eval_many(run_args, models[:max(1, min(len(models), n_groups))], spec_list, a.seq_len, a.overlap, tokenizer or tok,
                                     use_tqdm=False, return_device=not pcie, **({"group_sizes": group_sizes} if group_sizes and R > 1 else {}))
        if pcie:   # the reference's contract: np.float32 [T_ds, V+1] back on the host (lib.py:640), decoded from there
            return [decoder.ids(torch.from_numpy(o).to(dev)) for o in outs]
        return [decoder.ids(o) for o in outs]

    def timed(spec_list, pcie=None, **kw):
        if pcie:                                      # host-resident inputs are prepared outside the timed region
            spec_list = [sp if not sp.is_cuda else sp.cpu().pin_memory() for sp in spec_list]
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        run_many(spec_list, pcie=pcie, **kw)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t

    specs = [one_step(i) for i in range(a.warmup + a.steps)]
    if a.blank_bias < 0:  # shape the seeded model so pseudo-labels have a speech-like token rate (outside the timed region)
        from dynamic_asr_eval_amd.synthetic_weights import calibrate_blank_bias
        model.active = 1                       # the calibration forwards one window through replica 0
        a.blank_bias = calibrate_blank_bias(model, specs[0][:, :, :a.seq_len].contiguous().to(dev))
        model.active = R
    else:
        model.P["decoder.ff.bias"][-1] += a.blank_bias
    if R > 1:
        model.load_state_dict(model.state_dict())   # the calibrated blank bias into every replica
    for m in models[1:]:                       # every chain starts from the same weights
        m.flat_params.copy_(model.flat_params)
    progress("models built, calibrated; prewarm")
    if a.prewarm_s > 0:     # untimed, before the W warm-up steps: the workload itself, until the GPU has been under load for prewarm_s
        t_end = time.perf_counter() + a.prewarm_s
        while time.perf_counter() < t_end:
            run_many(specs[:1] * (n_chains * R))
        if R > 1 and a.graphs:
            # lib.lockstep_group_sizes may cut the K recordings into groups SMALLER than R (20 on 2 chains: 4 4 3 3 3 3); a partial group's replica
            # range has graphs of its own (model.py::_graph_pool), captured the second time it is seen — here, on every chain, not inside the timed region
            for n in sorted({sz for sz in lib.lockstep_group_sizes(a.steps, R, n_chains) if sz != R}, reverse=True):
                for _ in range(2):
                    run_many(specs[:1] * (n * n_chains), group_sizes=[n] * n_chains)
                    progress(f"prewarmed groups of {n}")
    progress("warm-up steps")
    if a.warmup:
        run_many(specs[:a.warmup])              # W untimed steps (every chain's stream and workspace was already warmed by the prewarm)
    ddist.barrier()
    torch.cuda.synchronize(dev)
    # Live roofline sampling.  Graph replays cannot be timed per launch, so every WEVERY-th window step of the timed region
    # runs eagerly with a HIP-event pair around every 4th GEMM launch.  Sampled steps alternate between "shared" (the other
    # chains keep the GPU busy: the duration a launch sees in this job, what rocprofv3 of this command averages) and
    # "exclusive" (the device is drained around the step: the kernel's own duration, what rocprofv3 --chains 1 averages).
    WEVERY = max(2, a.sample_every)
    ops.gemm_profile_start(every=4, window_every=WEVERY if a.graphs else 0)
    progress(f"timed region: {a.steps} steps")
    t0 = time.perf_counter()
    hyps = run_many(specs[a.warmup:a.warmup + a.steps])
    torch.cuda.synchronize(dev)
    ddist.barrier()
    dt = time.perf_counter() - t0
    progress(f"timed region done in {dt:.1f} s; side measurements")
    prof = ops.gemm_profile_stop()
    if os.environ.get("DYN_DEBUG_HOST"):
        print(f"[host] wall {dt:.3f} s, blocked on pseudo-label ids {lib.HOST_WAIT[0]:.3f} s (incl. warm-up)", file=sys.stderr)
    dt = ddist.max_over_ranks(dt)
    # the path's one collective (outside the timed region, as in the reference harness): hypothesis token counts over RCCL
    tokens_total = ddist.all_reduce_counts((sum(len(h) for h in hyps), len(hyps), 0, 0))
    # side measurements on rank 0 at N = 1 (untimed as far as `value` goes): the model's own collapsing pseudo-labels, and the
    # boundary-faithful rate (host spectrogram in, numpy log-probs out, decoded from the host copy: reference lib.py:549,640)
    side = {}
    n_side = n_chains * R if a.side_steps < 0 else a.side_steps
    if world == 1 and n_side > 0:
        sl = specs[a.warmup:a.warmup + min(n_side, a.steps)]
        if a.label_tokens > 0:
            side["value_degenerate_labels"] = round(a.seconds * len(sl) / timed(sl, tokenizer=plain_tok), 3)
            progress("side: degenerate labels done")
        if not a.pcie:
            side["value_boundary"] = round(a.seconds * len(sl) / timed(sl, pcie=1), 3)
            progress("side: boundary done")
        if not a.online:   # the mode of the reference's only published timing (`online=True`: no final pass, timeit_earnings22.sh:1)
            side["value_online"] = round(a.seconds * len(sl) / timed(sl, online=1), 3)
        side["side_sample"] = f"{len(sl)} recordings each, same chains, after the timed region"

    if rank == 0:
        audio_s = a.seconds * a.steps * world
        def tfs(k):
            return prof[k]["flops"] / (prof[k]["ms"] * 1e-3) / 1e12 if prof and prof[k]["ms"] > 0 else None
        shared, excl = tfs("shared"), tfs("exclusive")
        if n_chains == 1 and shared is not None and excl is not None:   # one chain: both kinds are the same measurement
            fl = prof["shared"]["flops"] + prof["exclusive"]["flops"]
            shared = excl = fl / ((prof["shared"]["ms"] + prof["exclusive"]["ms"]) * 1e-3) / 1e12
        achieved = excl if excl is not None else shared
        out = {
            "metric": "audio-sec/s dynamic-eval (fwd+1 adapt step)", "value": round(audio_s / dt, 3), "unit": "audio-s/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"run_dynamic_eval_full: {a.seconds / 3600:g} h Earnings-22-shape recording per step ({n_frames} frames, seq_len "
                                   f"{a.seq_len}, overlap {a.overlap}, {'online' if a.online else 'offline'}, MADGRAD lr {a.lr}, "
                                   "6 freq masks <=34), SCConformerXL 6x768 V+1=4096 seeded weights",
                       "recording_seconds": a.seconds, "windows_per_recording": len(lib.prepare_chunks(specs[0], a.seq_len, a.overlap)[1]),
                       "sharding": f"{world} ranks x {a.steps} recordings, no data-path collective",
                       "chains_per_gpu": n_chains, "lockstep_group": R, "hip_graphs": bool(a.graphs), "pcie_inclusive": bool(a.pcie), "prewarm_s": a.prewarm_s,
                       "blank_bias": round(a.blank_bias, 4), "hyp_tokens_per_recording": [len(h) for h in hyps]},
            "roofline": {"bound": "mfma", "achieved": None if achieved is None else round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic(), "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                         "algorithmic_bytes_per_launch": round(prof["bytes"] / prof["calls"]) if prof and prof["calls"] else None,
                         "flop_per_launch": round(prof["flops"] / prof["calls"]) if prof and prof["calls"] else None,
                         "achieved_shared": None if shared is None else round(shared, 2),
                         # whole-job check: all matrix-core flops of the timed region (GEMM launches + the fused attention of the final pass) / its wall
                         # time (every kernel, sync and gap included)
                         # extrapolated from the eagerly run (sampled) steps: only quoted when there are enough of them (>= 30; the driver's 20-step form has ~65)
                         "job_gemm_tflops": round((prof["flops"] + prof["attn_flops"]) * (WEVERY if a.graphs else 1) / dt / 1e12, 2)
                                            if prof and (not a.graphs or prof.get("sampled_steps", 0) >= 30) else None,
                         "gemm_launches": prof["calls"] if prof else 0,
                         "sampled_launches": (prof["exclusive"]["sampled"], prof["shared"]["sampled"]) if prof else 0,
                         "gemm_tflop_per_step": round(prof["flops"] * (WEVERY if a.graphs else 1) / a.steps / 1e12, 2)
                                                if prof and (not a.graphs or prof.get("sampled_steps", 0) >= 30) else None,
                         "sampled_steps": prof.get("sampled_steps") if prof else None,
                         "sampling": f"every 4th GEMM launch of every {WEVERY}th window step (those steps run eagerly; the rest replay "
                                     "hipGraphs); `achieved` = launches timed with the other chains drained (kernel's own duration), "
                                     "`achieved_shared` = launches timed while the other chains share the GPU"
                                     if a.graphs else "every 4th GEMM launch"},
            "hyp_tokens_total": int(tokens_total[0]),
        }
        out.update(side)
        out["config"]["label_tokens_per_window"] = a.label_tokens
        side_model = model
        if R > 1 and world == 1 and (a.side_workloads or not a.no_cpu_baseline):   # the side legs drive one recording at a time: a plain model
            side_model = SCConformerXL(vocab_size=a.vocab, device=dev)
            side_model.load_state_dict(model.state_dict())
        if world == 1 and a.side_workloads:
            progress("side: online done; other workloads")
            out["other_workloads"] = other_workloads(a, side_model, dev)
        if world == 1 and not a.no_cpu_baseline:
            progress("cpu baseline + parity")
            for m in models + [side_model]:                            # the parity leg runs on the bench's own (restored) weights
                m.use_graphs = bool(a.graphs)
            out["cpu_baseline"], out["parity"] = cpu_baseline(a, model, dev)       # parity through the path that was timed (the group model when R > 1)
            out["parity"]["path"] = f"lockstep group of {R} (the same recording on every replica; replicas agree to rounding: replica_spread_offline_online)" if R > 1 else "one recording per model"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
