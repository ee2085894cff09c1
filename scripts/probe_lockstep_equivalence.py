"""r04: lockstep group vs one recording at a time on the small model, the sequence of tests/test_model_gpu.py::test_lockstep_dynamic_eval_... printed run by
run (max |dlogp| per recording; 0.0 = bit-identical): equal lengths, partial groups, different lengths, two epochs, graphs on / off, repeated calls.
It found the shared-graph-pool hazard that `model._graph_pool` (one private pool per replica range) removes."""
import sys, argparse, numpy as np, torch
sys.path.insert(0, '.')
from dynamic_asr_eval_amd import lib
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
from oracle import dynamic_eval_ref as R_
from oracle.conformer_ref import SCConformerXLRef
SMALL = dict(n_layers=2, d_model=256, n_heads=2, head_dim=128, subsampling_conv_channels=64)
cuda = torch.device('cuda:0')
def _args(**kw):
    a = argparse.Namespace(); a.config = {'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {}}; a.__dict__.update(kw); return a
def _masks_for(keys, F, u, seed):
    g = torch.Generator().manual_seed(seed); return {k: (R_.draw_masks(3, 12, F, g), ([], [])) for k in keys}
R, vocab = 3, 128
ref = SCConformerXLRef(SMALL, vocab_size=vocab, seed=5, blank_bias=1.5)
single = SCConformerXL(SMALL, vocab_size=vocab, device=cuda); single.load_state_dict(ref.state_dict())
grp = SCConformerXL(SMALL, vocab_size=vocab, device=cuda, group=R); grp.load_state_dict(ref.state_dict())
tok = SyntheticTokenizer(vocab)
g = torch.Generator().manual_seed(21)
specs = [torch.randn(1, 80, 1500, generator=g) for _ in range(R)]
_, keys = R_.prepare_chunks(specs[0], 512, 256)
masks = [_masks_for(keys, 80, None, seed=30 + r) for r in range(R)]
for online in (False, True):
    want = []
    for r in range(R):
        a = _args(optim_lr=1e-4, epochs=1, shuffle=False, online=online, spec_augment_fixed_masks=masks[r], quiet=True)
        want.append(lib.dynamic_eval(a, single, specs[r], 512, 256, tok, use_tqdm=False, return_params=True))
    for n_rec in (R, 2):
        a = _args(optim_lr=1e-4, epochs=1, shuffle=False, online=online, spec_augment_fixed_masks=masks[:n_rec], quiet=True)
        got = lib.dynamic_eval_lockstep(a, grp, specs[:n_rec], 512, 256, tok, use_tqdm=False, return_params=True)
        print('equal', online, n_rec, [float(np.abs(got[r][0] - want[r][0]).max()) for r in range(n_rec)], flush=True)
lens = (1100, 1500, 1360)
specs2 = [torch.randn(1, 80, n, generator=g) for n in lens]
masks2 = [_masks_for(range(0, 2048, 256), 80, None, seed=50 + r) for r in range(R)]
for online in (False, True):
    want = []
    for r in range(R):
        a = _args(optim_lr=1e-4, epochs=1, shuffle=False, online=online, spec_augment_fixed_masks=masks2[r], quiet=True)
        want.append(lib.dynamic_eval(a, single, specs2[r], 512, 256, tok, use_tqdm=False, return_params=True))
    a = _args(optim_lr=1e-4, epochs=1, shuffle=False, online=online, spec_augment_fixed_masks=masks2, quiet=True)
    got = lib.dynamic_eval_lockstep(a, grp, specs2, 512, 256, tok, use_tqdm=False, return_params=True)
    print('unequal', online, [float(np.abs(got[r][0] - want[r][0]).max()) for r in range(R)], flush=True)
for graphs in (True, False, True):
    want = []
    for r in range(R):
        a = _args(optim_lr=1e-4, epochs=2, shuffle=False, spec_augment_fixed_masks=masks2[r], quiet=True, use_graphs=graphs)
        want.append(lib.dynamic_eval(a, single, specs2[r], 512, 256, tok, use_tqdm=False))
    a = _args(optim_lr=1e-4, epochs=2, shuffle=False, spec_augment_fixed_masks=masks2, quiet=True, use_graphs=graphs)
    got = lib.dynamic_eval_lockstep(a, grp, specs2, 512, 256, tok, use_tqdm=False)
    print('2 epochs graphs', graphs, [float(np.abs(got[r] - want[r]).max()) for r in range(R)], flush=True)
    a = _args(optim_lr=1e-4, epochs=2, shuffle=False, spec_augment_fixed_masks=masks2, quiet=True, use_graphs=graphs)
    got2 = lib.dynamic_eval_lockstep(a, grp, specs2, 512, 256, tok, use_tqdm=False)
    print('   again', [float(np.abs(got2[r] - want[r]).max()) for r in range(R)], [float(np.abs(got2[r] - got[r]).max()) for r in range(R)], flush=True)
    want2 = [lib.dynamic_eval(_args(optim_lr=1e-4, epochs=2, shuffle=False, spec_augment_fixed_masks=masks2[r], quiet=True, use_graphs=graphs), single, specs2[r], 512, 256, tok, use_tqdm=False) for r in range(R)]
    print('   single again vs single', [float(np.abs(want2[r] - want[r]).max()) for r in range(R)], flush=True)
