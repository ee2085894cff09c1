"""GreedyCTCDecoder with the reference's call signature — `GreedyCTCDecoder(tokenizer=..., blank_id=...)` then
`decoder(log_probs[T, C]) -> str` (reference lcasr/lib.py:498,559,565; run_dynamic_eval_full.py:53,100) — but the
argmax / collapse / blank-drop runs on the GPU (dyn_ctc_greedy) and only the surviving token ids cross PCIe."""
import torch

from . import ops


class GreedyCTCDecoder:
    def __init__(self, tokenizer, blank_id, device=None):
        self.tokenizer = tokenizer
        self.blank_id = int(blank_id)
        self.device = device

    def ids(self, log_probs):
        """[T, C] (or [B, T, C]) log-probabilities -> list (or list of lists) of token ids."""
        lp = torch.as_tensor(log_probs)
        if not lp.is_cuda:
            dev = self.device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
            if dev is None:
                raise ops.DynError("GreedyCTCDecoder: no GPU available and there is no CPU decode path")
            lp = lp.to(dev, torch.float32)
        lp = lp.contiguous()
        single = lp.dim() == 2
        ids, n = ops.ctc_greedy(lp, self.blank_id)
        n = n.cpu().tolist()
        ids = ids.cpu()
        out = [ids[b, :n[b]].tolist() for b in range(len(n))]
        return out[0] if single else out

    def __call__(self, log_probs):
        ids = self.ids(log_probs)
        if ids and isinstance(ids[0], list):
            return [self.tokenizer.decode(i) for i in ids]
        return self.tokenizer.decode(ids)
