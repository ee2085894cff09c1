"""Autotunes dyn_gemm_f32 for every GEMM shape the dynamic-eval step issues, ON the MI355X, on the real operands
(strides included), and writes dynamic-asr-eval_amd/csrc/gemm_tuned.inc (static table consulted by the planner).
Usage (GPU box):  python scripts/tune_gemm.py gpurun_out/gemm_tuned.inc   then copy the file into csrc/ and rebuild."""
import sys, os, argparse, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops, lib
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
from dynamic_asr_eval_amd.datasets import synthetic_spec
from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_tuned.inc"
SHARED = 3   # second table: the same GEMM launched on 3 streams at once (what a launch meets when recording chains share the GPU)
dev = torch.device("cuda:0")
model = SCConformerXL(vocab_size=4095, device=dev)
init_synthetic(model, seed=0, blank_bias=1.34)
tok = SyntheticTokenizer(4095)
args = argparse.Namespace(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {}},
                          optim_lr=9e-5, epochs=1, shuffle=False, quiet=True, use_graphs=False, spec_augment_n_freq_masks=6, spec_augment_freq_mask_param=34)
# 1 h shape: full windows + the short last window (15936 frames), final pass batched by 4
spec = synthetic_spec(16384 + 8 * 2048 - 448, seed=1).to(dev)
orig = ops.gemm
best = {}
log = []

def timeit(fn, n=8):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

streams = [torch.cuda.Stream() for _ in range(SHARED)]
best_shared = {}

def timeit_shared(call, n=6):
    """call(i) launches the GEMM into scratch i on the current stream; SHARED streams run it concurrently."""
    for i, st in enumerate(streams):
        with torch.cuda.stream(st): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in streams: st.wait_event(e0)
    for _ in range(n):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st): call(i)
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * SHARED)


def tuned(a, b, c, **kw):
    key = (bool(kw.get('trans_a', False)), bool(kw.get('trans_b', False)), kw['M'], kw['N'], kw['K'], kw.get('nb1', 1) * kw.get('nb2', 1))
    if key not in best and kw.get('split_k', 0) == 0:
        scratch = c.clone()
        K = kw['K']
        cands = []
        for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
            if (kw['M'] <= 64 and bm == 128) or (kw['N'] <= 64 and bn == 128):
                continue
            for f in (1, 2, 3, 4, 6, 8):
                cands.append((bm, bn, 1, f))
            for s in (2, 3, 4, 6, 8, 12):
                if K // s >= 128:
                    cands.append((bm, bn, s, 1))
        res = []
        kw2 = dict(kw)
        for bm, bn, s, f in cands:
            kw2['split_k'] = s; kw2['force'] = (bm, bn, f)
            try:
                ms = timeit(lambda: orig(a, b, scratch, **kw2))
            except Exception as e:
                continue
            res.append((ms, bm, bn, s, f))
        res.sort()
        kw0 = dict(kw)
        base = timeit(lambda: orig(a, b, scratch, **kw0))
        best[key] = res[0]
        # throughput table: candidates without tail slicing, timed under sharing
        scr = [c.clone() for _ in range(SHARED)]
        res2 = []
        for bm, bn, s_, f in cands:
            if f != 1:
                continue
            kw3 = dict(kw); kw3['split_k'] = s_; kw3['force'] = (bm, bn, 1)
            try:
                ms = timeit_shared(lambda i: orig(a, b, scr[i], **kw3))
            except Exception:
                continue
            res2.append((ms, bm, bn, s_, 1))
        res2.sort()
        best_shared[key] = res2[0]
        del scr
        fl = 2.0 * kw['M'] * kw['N'] * kw['K'] * key[5]
        log.append(f"{'T' if key[0] else 'N'}{'T' if key[1] else 'N'} M={key[2]} N={key[3]} K={key[4]} nb={key[5]}: best {res[0][1]}x{res[0][2]} s={res[0][3]} f={res[0][4]} "
                   f"{res[0][0]*1e3:.1f}us {fl/res[0][0]/1e9:.1f} TF/s | planner {base*1e3:.1f}us {fl/base/1e9:.1f} TF/s | 2nd {res[1][1]}x{res[1][2]} s={res[1][3]} f={res[1][4]} {res[1][0]*1e3:.1f}us")
        log[-1] += f" || shared x{SHARED}: best {res2[0][1]}x{res2[0][2]} s={res2[0][3]} {res2[0][0]*1e3:.1f}us/launch, 2nd {res2[1][1]}x{res2[1][2]} s={res2[1][3]} {res2[1][0]*1e3:.1f}us"
        print(log[-1], flush=True)
        del scratch
    return orig(a, b, c, **kw)

ops.gemm = tuned
lib.dynamic_eval(args, model, spec, 16384, 14336, tok, use_tqdm=False, return_device=True)
torch.cuda.synchronize()
with open(out_path, "w") as f:
    f.write("// generated by scripts/tune_gemm.py on an MI355X (gfx950): best (tile, split-K, tail slices) per GEMM shape of the\n"
            "// dynamic-eval step, timed with HIP events on the real operands.  {ta, tb, M, N, K, batch, bm, bn, split, tail}\n")
    f.write("static const Tuned kTuned[] = {\n")
    for key, (ms, bm, bn, s, fs) in sorted(best.items()):
        f.write(f"    {{{int(key[0])}, {int(key[1])}, {key[2]}, {key[3]}, {key[4]}, {key[5]}, {bm}, {bn}, {s}, {fs}}},\n")
    f.write("};\n")
    f.write(f"static const int kNumTuned = {len(best)};\n")
    f.write(f"// same shapes with {SHARED} streams launching the same GEMM concurrently: least time per launch under sharing, no tail slicing\n")
    f.write("static const Tuned kTunedShared[] = {\n")
    for key, (ms, bm, bn, s_, fs) in sorted(best_shared.items()):
        f.write(f"    {{{int(key[0])}, {int(key[1])}, {key[2]}, {key[3]}, {key[4]}, {key[5]}, {bm}, {bn}, {s_}, {fs}}},\n")
    f.write("};\n")
    f.write(f"static const int kNumTunedShared = {len(best_shared)};\n")
print(f"wrote {len(best)} entries to {out_path}")
