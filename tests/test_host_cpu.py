"""CPU suite (-m "not gpu"): host logic, the C-ABI surface, and the oracle against values taken from the reference's
own files.  No compute call on the HIP library is made here (there is no GPU in this container)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------------ C-ABI surface
def test_library_loads_and_exports_every_declared_symbol():
    from dynamic_asr_eval_amd import _lib
    lib = _lib.load()
    names = _lib.exported_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dyneval.h but not exported by libdyneval_hip.so"
    assert lib.dyn_arch() == b"gfx950"
    assert lib.dyn_version().startswith(b"dyneval-hip")


def test_abi_argument_errors_are_codes_not_crashes():
    """Argument validation runs on the host before any launch, so it is checkable without a GPU."""
    from dynamic_asr_eval_amd import _lib
    lib = _lib.load()
    assert lib.dyn_silu_fwd(None, None, 16, None) == -1          # DYN_E_ARG
    assert b"dyn_silu_fwd" in lib.dyn_last_error()
    assert lib.dyn_glu_fwd(1, 1, 4, 6, None) == -1               # C % 4 != 0
    assert lib.dyn_layernorm_fwd(1, 1, 1, 1, 1, 1, 4, 100, 1e-5, None) == -1   # C % 256 != 0
    assert lib.dyn_ctc_loss(1, 8, 1, 5, 5, 40, 1, 4, 1, 1, 7, 0, 1.0, 1, None, None, 5, 40, None, 0, None) == -1  # blank >= C
    assert lib.dyn_ctc_loss_workspace_bytes(2048, 1, 500) > 2 * 2048 * 1001 * 4
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = 768, 768, 4096, 1, 1
    assert lib.dyn_gemm_f32_workspace_bytes(ctypes.byref(d)) > 0  # deep-K wgrad shape gets a split-K plan
    d.M, d.N, d.K = 4096, 3072, 768
    assert lib.dyn_gemm_f32_workspace_bytes(ctypes.byref(d)) == 0


def test_product_path_fails_loudly_without_gpu():
    from dynamic_asr_eval_amd import ops
    from dynamic_asr_eval_amd._lib import DynError
    with pytest.raises(DynError):
        ops.silu(torch.zeros(8))
    from dynamic_asr_eval_amd.model import SCConformerXL
    with pytest.raises(DynError):
        SCConformerXL(device="cpu")
    from dynamic_asr_eval_amd.decoding import GreedyCTCDecoder
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    if not torch.cuda.is_available():
        with pytest.raises(DynError):
            GreedyCTCDecoder(SyntheticTokenizer(8), 8)(torch.zeros(4, 9))


# ------------------------------------------------------------------------------------------------ windowing / stitch plan
@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_prepare_chunks_matches_reference_counts(impl):
    """Window counts stated by the reference material: 1 h (360000 frames) at 16384/14336 -> 169 windows, last 15936
    (from reference lcasr/lib.py:128-145); 415990 frames -> 197 (timeit_earnings22.sh:6 recording)."""
    if impl == "product":
        from dynamic_asr_eval_amd.lib import prepare_chunks
    else:
        from oracle.dynamic_eval_ref import prepare_chunks
    gold = json.load(open(os.path.join(GOLDEN, "prepare_chunks.json")))
    for case in gold["cases"]:
        spec = torch.empty(1, 1, case["spec_n"])
        data, keys = prepare_chunks(spec, case["seq_len"], case["overlap"])
        assert len(keys) == case["n_windows"], case
        assert keys[:3] == case["first_keys"] and keys[-1] == case["last_key"]
        assert data[keys[-1]].shape[-1] == case["last_len"]
        assert all(data[k].data_ptr() == spec[:, :, k:].data_ptr() for k in keys)  # views, not copies


def test_apply_args_flag_surface_and_kwargs():
    from dynamic_asr_eval_amd import lib
    p = argparse.ArgumentParser()
    args = lib.apply_args(p, ["-ds", "-epochs", "2", "-seq", "2048", "-o", "0", "-dfa", "-kwargs", "optim_lr=9e-6",
                              "spec_augment_n_freq_masks=6", "online=True"])
    assert args.seq_len == 2048 and args.overlap == 0 and args.epochs == 2 and args.shuffle is False
    assert args.optim_lr == 9e-6 and args.online is True and args.disable_flash_attention
    assert lib.get_lr_args_from_args(args) == {"lr": 9e-6}
    assert lib.get_lr_args_from_args(argparse.Namespace()) == {"lr": 9e-5}            # reference lib.py:124
    sa = lib.get_specaugment_config_from_args(args)
    assert sa == {"n_time_masks": 0, "n_freq_masks": 6, "freq_mask_param": 42, "time_mask_param": -1, "min_p": 0.05,
                  "zero_masking": False}                                                # reference lib.py:104-111
    assert lib.get_cutout_params_from_args(args, 2048)["num_rectangles"] == 0
    assert lib.get_frame_shuffle_config_from_args(args) == {"time_dimension": False, "freq_dimension": False}


def test_optional_augmentations_refuse_instead_of_silently_skipping():
    from dynamic_asr_eval_amd import lib

    class M:
        device = torch.device("cuda:0")
    a = argparse.Namespace(config={"model": {"subsampling_factor": 8}, "audio_chunking": {"size": 1, "overlap": 0}, "training": {}},
                           lm_tta_beams=3)
    with pytest.raises(NotImplementedError):   # LM beam-search pseudo-labels need the un-vendored `lming` LM
        lib.dynamic_eval(a, M(), torch.zeros(1, 80, 10), 8, 0, None, beam_search_fn=object())


# ------------------------------------------------------------------------------------------------ host utilities
def test_wer_counts_and_rates():
    from dynamic_asr_eval_amd.wer import edit_counts, word_error_rate_detail, basic_normalize
    assert edit_counts(["a b c d", "x y"], ["a c d e", "x y z"]) == (1, 2, 0, 7)
    wer, words, ins, dele, sub = word_error_rate_detail(["the cat sat"], ["the cat sat on the mat"])
    assert (wer, words, ins, dele, sub) == (0.5, 6, 0.0, 0.5, 0.0)
    assert word_error_rate_detail(["a"], ["a"])[0] == 0.0
    assert basic_normalize("Hello,  WORLD! it's") == "hello world it's"
    rng = np.random.RandomState(0)

    def lev(h, r):
        d = list(range(len(h) + 1))
        for i in range(1, len(r) + 1):
            nd = [i] + [0] * len(h)
            for j in range(1, len(h) + 1):
                nd[j] = min(d[j - 1] + (h[j - 1] != r[i - 1]), d[j] + 1, nd[j - 1] + 1)
            d = nd
        return d[-1]
    from dynamic_asr_eval_amd.wer import _align
    for _ in range(200):
        h = rng.randint(0, 5, rng.randint(0, 14)).tolist(); r = rng.randint(0, 5, rng.randint(0, 14)).tolist()
        i, d, s = _align(h, r)
        assert i + d + s == lev(h, r) and len(h) - i == len(r) - d


def test_tokenizers_round_trip():
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer, load_sentencepiece
    t = SyntheticTokenizer(4095)
    ids = [0, 17, 4094, 17, 3]
    assert t.encode(t.decode(ids)) == ids and t.vocab_size() == 4095 and t.encode("") == []
    sp = load_sentencepiece(os.path.join(GOLDEN, "tokenizer_128.model"))   # the reference's only tokenizer asset
    assert sp.vocab_size() == 128
    text = "hello world this is a test"
    assert sp.decode(sp.encode(text)) == text


def test_specaugment_draw_rule_is_shared_with_oracle():
    from dynamic_asr_eval_amd.augment import SpecAugment, draw_masks
    from oracle.dynamic_eval_ref import draw_masks as draw_ref
    a = draw_masks(6, 34, 80, torch.Generator().manual_seed(3))
    b = draw_ref(6, 34, 80, torch.Generator().manual_seed(3))
    assert a == b and all(0 <= s and s + w <= 80 and w <= 34 for s, w in zip(*a))
    fm, tm = SpecAugment(n_freq_masks=6, freq_mask_param=34, n_time_masks=2, time_mask_param=-1, min_p=0.05).draw(80, 1000, torch.Generator().manual_seed(1))
    assert len(fm[0]) == 6 and len(tm[0]) == 2 and all(w <= 50 for w in tm[1])


def test_shard_longest_first_is_balanced_and_deterministic():
    from dynamic_asr_eval_amd.dist import shard_longest_first
    lens = [360000, 415990, 120000, 90000, 90000, 300000, 45000]
    bins = shard_longest_first(lens, 3)
    assert sorted(i for b in bins for i in b) == list(range(7))
    loads = [sum(lens[i] for i in b) for b in bins]
    assert max(loads) - min(loads) <= max(lens)
    assert bins == shard_longest_first(lens, 3)
    assert shard_longest_first(lens, 1) == [list(range(7))]
    assert shard_longest_first([5, 4], 4) == [[0], [1], [], []]


def test_gloo_world2_counter_allreduce_and_gather():
    """The N>1 path on CPU: 2 ranks, gloo, recordings sharded, the WER counters all-reduced, records gathered."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", script],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["counts"] == res["expected_counts"]
    assert res["ids"] == [f"rec{i}" for i in range(5)]
    assert res["max_elapsed"] == 2.0


# ------------------------------------------------------------------------------------------------ oracle pins
def test_oracle_ctc_greedy_and_stitch_toy():
    """3-window toy of the stitch arithmetic (reference lcasr/lib.py:615-629) worked by hand."""
    from oracle.dynamic_eval_ref import greedy_ctc_ids
    lp = torch.log(torch.tensor([[.1, .8, .1], [.1, .8, .1], [.1, .1, .8], [.7, .2, .1], [.1, .8, .1]]))
    assert greedy_ctc_ids(lp, blank_id=2) == [1, 0, 1]
    gold = json.load(open(os.path.join(GOLDEN, "stitch_toy.json")))
    seq_len, overlap, ds = gold["seq_len"], gold["overlap"], gold["downsample"]
    pos, cover = 0, {}
    for k, u_len in zip(gold["keys"], gold["u_lens"]):
        ds_len = u_len // ds
        ov = int(overlap / (u_len / ds_len))
        pos -= ov if k != 0 else 0
        for r in range(pos, pos + ds_len):
            cover[r] = cover.get(r, 0) + 1
        pos += ds_len
    assert [cover[r] for r in sorted(cover)] == gold["counts"]


def test_oracle_softdtw_matches_reference_recurrence_fixture():
    """Soft-DTW values and gradients for the reference's own self-check shape family (soft_dtw_cuda.py:382-428),
    generated by oracle/softdtw_ref.py (numpy fp64 restatement of soft_dtw_cuda.py:184-239) — see tests/golden/README.md."""
    from oracle.softdtw_ref import softdtw_forward_backward
    z = np.load(os.path.join(GOLDEN, "softdtw_17x15x2.npz"))
    R, E = softdtw_forward_backward(z["D"], float(z["gamma"]), 0.0)
    assert np.allclose(R, z["value"], rtol=0, atol=1e-12)
    assert np.allclose(E, z["grad"], rtol=0, atol=1e-12)
    # hand-checkable case: 1x1 -> value = D, grad = 1; and gamma -> 0 approaches hard DTW
    R1, E1 = softdtw_forward_backward(np.array([[[3.5]]]), 1.0, 0.0)
    assert R1[0] == 3.5 and E1[0, 0, 0] == 1.0
    D = np.array([[[1., 9., 9.], [9., 2., 9.], [9., 9., 3.]]])
    Rh, _ = softdtw_forward_backward(D, 1e-3, 0.0)
    assert abs(Rh[0] - 6.0) < 1e-2


def test_oracle_model_and_madgrad_are_self_consistent():
    from oracle.conformer_ref import SCConformerXLRef
    from oracle.madgrad_ref import MADGRAD
    from dynamic_asr_eval_amd.model import param_spec, make_config
    cfg = dict(n_layers=1, d_model=256, n_heads=2, head_dim=64, subsampling_conv_channels=32)
    m = SCConformerXLRef(cfg, vocab_size=16, seed=0)
    assert [(n, tuple(p.shape)) for n, p in m.named_parameters()] == [(n, tuple(s)) for n, s in param_spec(make_config(**cfg), 17)]
    out = m(audio_signal=torch.randn(2, 80, 64))["final_posteriors"]
    assert out.shape == (2, 8, 17) and torch.allclose(out.exp().sum(-1), torch.ones(2, 8), atol=1e-5)
    assert SCConformerXLRef(vocab_size=4095).decoder.num_classes == 4096
    # MADGRAD, one step from zero state with momentum 0: p1 = p0 - lamb*g / (cbrt(lamb*g^2) + eps)
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0]))
    opt = MADGRAD([p], lr=0.1, momentum=0.0, eps=1e-6)
    p.grad = torch.tensor([0.5, 0.25]); opt.step()
    lamb = 0.1 + 1e-6
    exp = torch.tensor([1.0, -2.0]) - lamb * torch.tensor([0.5, 0.25]) / ((lamb * torch.tensor([0.25, 0.0625])).pow(1 / 3) + 1e-6)
    assert torch.allclose(p.detach(), exp, atol=1e-7)


def test_loo_disjoint_pairs_rule():
    """reference run_within_recording_loo_eval.py:116-121: chunk k covers [k, k + len_k); i and j pair up only if disjoint."""
    from dynamic_asr_eval_amd.run_within_recording_loo_eval import disjoint_pairs
    keys = [0, 256, 512, 768, 1024]
    lens = {0: 512, 256: 512, 512: 512, 768: 512, 1024: 376}
    v = disjoint_pairs(keys, lens)
    assert v[0] == [512, 768, 1024] and v[256] == [768, 1024] and v[512] == [0, 1024] and v[1024] == [0, 256, 512]
    assert disjoint_pairs([0, 100], {0: 300, 100: 300}) == {0: [], 100: []}


def test_stm_parsing_follows_the_reference_rules():
    """reference lcasr/tedlium/run.py:30-51."""
    from dynamic_asr_eval_amd.datasets import proc_stm_lines, get_text_and_audio_synthetic_tedlium
    lines = ["t 1 s 0.00 2.50 <o> hello  world it 's fine", "short line", "t 1 s 2.50 4.00 <o> ignore_time_segment_in_scoring",
             "t 1 s 4.00 6.25 <o> we 're   back"]
    text, keep, remove = proc_stm_lines(lines)
    assert text == "hello world it's fine we're back"
    assert keep == [{'start': 0.0, 'end': 2.5}, {'start': 4.0, 'end': 6.25}] and remove == [{'start': 2.5, 'end': 4.0}]
    recs = get_text_and_audio_synthetic_tedlium('test', durations_s=[60.0])
    _, _, rem = proc_stm_lines(recs[0]['stm'])
    assert len(rem) >= 1 and recs[0]['frames'] == 6001


def test_oracle_reproduces_the_committed_kernel_fixtures():
    """tests/golden/{ctc,adam,madgrad}*.npz are what the GPU tests compare against: check here that the CPU side (torch ops
    the reference calls / the oracle restatement) still produces those numbers in this environment."""
    d = np.load(os.path.join(GOLDEN, "ctc_64x10x129.npz"))
    lp = torch.from_numpy(d["log_probs"]).requires_grad_(True)
    loss = torch.nn.CTCLoss(blank=128, reduction='sum')(lp, torch.from_numpy(d["targets"]).long(), torch.from_numpy(d["input_lengths"]).long(),
                                                       torch.from_numpy(d["target_lengths"]).long())
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 1e-4 and np.abs(lp.grad.numpy() - d["grad"]).max() < 1e-6
    from oracle.madgrad_ref import MADGRAD
    for name, make in (("adam_3step.npz", lambda p, lr: torch.optim.Adam([p], lr=lr)), ("madgrad_3step.npz", lambda p, lr: MADGRAD([p], lr=lr))):
        d = np.load(os.path.join(GOLDEN, name))
        p = torch.nn.Parameter(torch.from_numpy(d["p0"]).clone())
        opt = make(p, float(d["lr"]))
        for k in range(3):
            p.grad = torch.from_numpy(d["grads"][k]).clone(); opt.step()
            assert np.abs(p.detach().numpy() - d["params"][k]).max() < 1e-6, (name, k)


def test_oracle_loop_reproduces_the_committed_trace():
    from oracle import dynamic_eval_ref as R
    from oracle.conformer_ref import SCConformerXLRef
    from oracle.madgrad_ref import MADGRAD
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    d = np.load(os.path.join(GOLDEN, "dyneval_trace.npz"))
    cfg = dict(n_layers=2, d_model=256, n_heads=2, head_dim=128, subsampling_conv_channels=64)
    ref = SCConformerXLRef(cfg, vocab_size=128, seed=int(d["model_seed"]), blank_bias=float(d["blank_bias"]))
    masks = {int(k): ((list(map(int, s)), list(map(int, w))), ([], [])) for k, s, w in zip(d["keys"], d["mask_starts"], d["mask_widths"])}
    out = R.dynamic_eval_ref(ref, torch.from_numpy(d["spec"]), 512, 256, SyntheticTokenizer(128), MADGRAD, {'lr': float(d["lr"])}, {},
                             epochs=1, shuffle=False, online=False, fixed_masks=masks)
    assert np.abs(out - d["logits"]).max() < 1e-4 and np.array_equal(out.argmax(-1).astype(np.int32), d["argmax"])


def test_speaker_manifest_loader_on_the_reference_manifest():
    """reference run_cross_speaker_gender_tedlium.py:31-39 on the manifest the reference ships (15 + 15 talks)."""
    from dynamic_asr_eval_amd.run_cross_speaker_gender_tedlium import DEFAULT_SPEAKER_MANIFEST, load_speaker_manifest
    manifest, gender = load_speaker_manifest(DEFAULT_SPEAKER_MANIFEST)
    assert len(manifest['female']) == 15 and len(manifest['male']) == 15 and len(gender) == 30
    assert gender['JaneMcGonigal_2010.sph'] == 'F' and sorted(set(gender.values())) == ['F', 'M']


def test_bench_chain_count_avoids_a_lone_last_recording():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert [m.pick_chains(3, k) for k in (1, 2, 3, 4, 5, 6, 10)] == [1, 2, 3, 2, 3, 3, 2]
    assert m.pick_chains(1, 7) == 1


def test_product_package_never_imports_the_oracle():
    """The oracle is a checker: no module of the product package (or bench.py outside its cpu_baseline leg) may import it."""
    import ast, glob
    pkg = os.path.join(ROOT, "dynamic-asr-eval_amd")
    for path in glob.glob(os.path.join(pkg, "*.py")):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{path} imports the oracle"
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert not uses or fn.name.startswith("cpu_baseline"), f"bench.py:{fn.name} imports the oracle outside the cpu_baseline leg"
    # and the product fails loudly, not silently, when the HIP library is absent
    from dynamic_asr_eval_amd import _lib
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['DYN_LIB_PATH'] = '/nonexistent/libdyneval_hip.so'\n"
            "from dynamic_asr_eval_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.DynError as e:\n    print('DynError', 'no CPU fallback' in str(e))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.stdout.strip() == "DynError True", out.stdout + out.stderr


def test_bench_gpus2_self_launches_its_ranks_and_rendezvous():
    """`python bench.py --gpus 2` is the driver's command form: with no torchrun environment bench.py must start the ranks itself
    (before any GPU call) and relay rank 0's JSON line.  --rendezvous_only stops after the process group, the barrier and the two
    collectives of the bench (no GPU needed): 2 gloo ranks here."""
    env = dict(os.environ, DYN_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous_only"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res == {"rendezvous": True, "n_gpus": 2, "ranks_seen": 2, "rank_sum": 1, "max_rank": 1.0}


def test_one_rank_group_runs_the_collectives_through_the_backend():
    """dist.init(force=True) creates the one-rank group; the helpers then go through the backend instead of short-circuiting
    (gloo here; tests/test_dist_gpu.py does the same through RCCL on the GPU box)."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import torch.distributed as dist\n"
            "from dynamic_asr_eval_amd import dist as ddist\n"
            "assert ddist.all_reduce_counts((1, 2, 3, 4)) == (1, 2, 3, 4) and not dist.is_initialized()\n"
            "ddist.init(backend='gloo', force=True); assert dist.is_initialized() and dist.get_world_size() == 1\n"
            "assert ddist.all_reduce_counts((1, 2, 3, 4)) == (1, 2, 3, 4) and ddist.max_over_ranks(1.5) == 1.5\n"
            "assert [r['index'] for r in ddist.gather_records([{'index': 2}, {'index': 1}])] == [1, 2]\n"
            "ddist.barrier(); ddist.shutdown(); assert not dist.is_initialized(); print('OK')\n" % ROOT)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


def test_grouped_wgrad_queue_is_sliced_for_deep_models():
    """8 weight-gradient descriptors per block + 1 exceed the 96 groups of one dyn_gemm_f32_grouped launch from 12 blocks
    on; ops.gemm_grouped must cut the queue into launches of <= MAX_GROUPS (checked on the host: the C call is replaced)."""
    from dynamic_asr_eval_amd import ops
    from dynamic_asr_eval_amd._lib import GemmDesc
    calls = []

    class FakeLib:
        def dyn_gemm_f32_grouped(self, arr, n, ws, ws_bytes, stream):
            calls.append(int(n))
            return 0
    descs = []
    for _ in range(8 * 12 + 1):
        d = GemmDesc(); d.M, d.N, d.K = 4, 4, 4
        descs.append(d)
    old = (ops._L, ops.workspace, ops._stream)
    ops._L, ops.workspace, ops._stream = (lambda: FakeLib()), (lambda *a, **k: torch.empty(1024, dtype=torch.uint8)), (lambda: 0)
    try:
        ops.gemm_grouped(descs)
    finally:
        ops._L, ops.workspace, ops._stream = old
    assert calls == [96, 1] and ops.MAX_GROUPS == 96


def test_fused_attention_grad_switch_is_validated_where_it_is_set():
    from dynamic_asr_eval_amd.model import _parse_fused_attn_grad
    assert [_parse_fused_attn_grad(v) for v in ("0", "1", "4096", 2048, " 8192 ")] == [0, 1, 4096, 2048, 8192]
    for bad in ("auto", "", "-1", "1.5"):
        with pytest.raises(ValueError):
            _parse_fused_attn_grad(bad)


def test_libm_restatement_is_bitwise_the_system_libm(tmp_path):
    """csrc/libm_f32.h (the expf / logf the CTC lattice kernels use, so that they round like torch's CPU CTC) compiled for the HOST against
    this machine's libm.so.6 — every 251st float bit pattern (17 M inputs incl. negatives, subnormals, inf, nan; the full 2^32 sweep was run
    once in the build container: 0 mismatches) plus the 2^16 patterns around each special boundary."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    hdr = os.path.join(ROOT, "dynamic-asr-eval_amd", "csrc", "libm_f32.h")
    src = tmp_path / "libm_check.cpp"
    src.write_text('#include "%s"\n' % hdr + r'''
#include <stdio.h>
#include <math.h>
using namespace dyn::glm;
static long check(uint32_t u, const double* tab) {
    float x = u2f(u); long bad = 0;
    float a = expf(x), b = expf_(x, tab), c = logf(x), d = logf_(x, tab);
    if (f2u(a) != f2u(b) && !(a != a && b != b)) bad++;
    if (f2u(c) != f2u(d) && !(c != c && d != d)) bad++;
    if (x <= 0 && !(x != x)) { float e = exp_nonpos(x, tab); if (f2u(a) != f2u(e)) bad++; }
    return bad;
}
int main() {
    double tab[TABLE_DOUBLES]; fill_table_host(tab);
    long bad = 0, n = 0;
    for (uint64_t u = 0; u < (1ull << 32); u += 251) { bad += check((uint32_t)u, tab); n++; }
    const uint32_t around[] = {0x00000000u, 0x00800000u, 0x3f800000u, 0x42b00000u, 0x42b17218u, 0x7f800000u, 0x80000000u, 0xc2aeac50u,
                               0xc2cff1b4u, 0xc2ce8ed0u, 0xff800000u, 0x3f330000u};
    for (uint32_t c : around) for (int64_t k = -32768; k < 32768; ++k) { bad += check((uint32_t)((int64_t)c + k), tab); n++; }
    printf("%ld inputs %ld mismatches\n", n, bad);
    return bad != 0;
}
''')
    exe = tmp_path / "libm_check"
    subprocess.run(["g++", "-O2", "-std=c++17", str(src), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_lockstep_group_sizes_balance_the_chains():
    """lib.lockstep_group_sizes: every recording in exactly one group, no group above R, and with several chains the group count is a multiple of
    the chain count (sizes within one of each other) so that the chains finish together."""
    from dynamic_asr_eval_amd.lib import lockstep_group_sizes as g
    assert g(20, 4, 2) == [4, 4, 3, 3, 3, 3] and g(8, 4, 2) == [4, 4] and g(6, 4, 2) == [3, 3] and g(4, 4, 2) == [4] and g(0, 4, 2) == []
    assert g(7, 3, 1) == [3, 2, 2] and g(5, 1, 3) == [1] * 5 and g(2, 4, 3) == [2]
    for n in range(1, 60):
        for R in (1, 2, 3, 4, 6):
            for c in (1, 2, 3, 4):
                s = g(n, R, c)
                assert sum(s) == n and max(s) <= R and min(s) >= 1 and max(s) - min(s) <= 1 and s == sorted(s, reverse=True)
                if c > 1 and len(s) > 1 and n >= c * 2:
                    assert len(s) % c == 0 or len(s) == n, (n, R, c, s)


def test_wav2vec2_length_buckets_host_arithmetic():
    """wav2vec2_model.Wav2Vec2ForCTC.conv_lengths / samples_for_frames / _bucket (host side of the hipGraph length buckets): the bucket's sample
    count holds every utterance that has at most the bucket's frames, the bucket itself has exactly its frame count at every conv layer the
    static shapes are built from, and samples_for_frames is the smallest such length."""
    from types import SimpleNamespace
    from dynamic_asr_eval_amd.wav2vec2_model import DEFAULT_CONFIG, Wav2Vec2ForCTC as W
    me = SimpleNamespace(cfg=dict(DEFAULT_CONFIG), bucket_frames=32)
    me.conv_lengths = lambda L: W.conv_lengths(me, L)
    me.samples_for_frames = lambda T: W.samples_for_frames(me, T)
    assert me.samples_for_frames(1) == 400 and me.conv_lengths(400)[-1] == 1 and me.conv_lengths(16000)[-1] == 49       # HF: 49 frames per second
    for T in (1, 2, 31, 32, 33, 448, 1600):
        L = me.samples_for_frames(T)
        assert me.conv_lengths(L)[-1] == T and me.conv_lengths(L - 1)[-1] == T - 1
    for L in list(range(400, 2400, 7)) + [10239, 10240, 10559, 10560, 10561, 131072, 480000, 480319]:
        T, Tb, Lb = W._bucket(me, L)
        assert T == me.conv_lengths(L)[-1] and Tb % 32 == 0 and Tb - 32 < T <= Tb and Lb >= L
        assert me.conv_lengths(Lb)[-1] == Tb and me.conv_lengths(Lb + 1)[-1] == Tb + 1
        assert all(v <= b for v, b in zip(me.conv_lengths(L), me.conv_lengths(Lb)))      # every layer's valid frames fit the static shape


def test_bf16x3_split_is_exact_on_the_host_emulation():
    """The operand split of the experimental bf16x3 GEMM (csrc/gemm_bf16x3.hip, emulated in scripts/emulate_bf16x3_kernel.py): every finite fp32
    value of magnitude >= 2^-95 is EXACTLY the sum of its three round-to-nearest-even bf16 terms (3 x 8 significant bits cover the 24), also for
    negative and power-of-two values; what the kernel drops is products, never operand bits.  (Below ~2^-102 the third term would be a bf16
    subnormal and the sum is off by < 2^-133 absolute: far under anything an activation or weight of this model carries.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("emu", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "emulate_bf16x3_kernel.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * s for s in (1e-6, 1.0, 1e6, 1e30)] +
                       [np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -90, 2.0 ** 100, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -65504.0], np.float32),
                        rng.integers(0x10000000, 0x7F000000, 200000, dtype=np.uint32).view(np.float32)])
    x = x[(x == 0) | (np.abs(x) >= 2.0 ** -95)]
    t0, t1, t2 = emu.split3(x)
    total = emu.bf16_value(t0).astype(np.float64) + emu.bf16_value(t1).astype(np.float64) + emu.bf16_value(t2).astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))
