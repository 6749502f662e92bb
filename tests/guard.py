"""Guard-band allocation for the op tests (the GPU box has no AddressSanitizer).

Every operand and output of a launch under test is a view into ONE poisoned arena, with a margin of poison on both sides:
  * the poison is a NaN both as fp32 (0x7fc07fc0) and as a pair of bf16 (0x7fc0, 0x7fc0), so a load that leaves its tensor and
    is CONSUMED turns the result into NaN -- a deterministic failure instead of a box-dependent one;
  * `check()` compares every margin with the poison pattern afterwards: a store that leaves its tensor is caught even when
    nothing reads the place again.
The arena is sized for the op-test geometries (a few MB)."""
import numpy as np
import torch

POISON = 0x7fc07fc0
MARGIN = 1 << 16            # bytes on each side of every tensor (beyond the largest tile overhang of the op tests)


class Arena:
    def __init__(self, nbytes=64 << 20, device="cuda"):
        self.words = torch.full((nbytes // 4,), POISON, dtype=torch.int32, device=device)
        self.off = MARGIN
        self.spans = []             # (begin, end) byte ranges handed out

    def reset(self):
        self.words.fill_(POISON)
        self.off = MARGIN
        self.spans = []

    def _carve(self, nbytes):
        begin = (self.off + 255) // 256 * 256
        end = begin + nbytes
        assert end + MARGIN <= self.words.numel() * 4, "arena too small"
        self.spans.append((begin, end))
        self.off = end + MARGIN
        return begin

    def empty(self, shape, dtype=torch.float32):
        """an UNINITIALISED tensor: its contents are poison (an output every element of which the launch must write)"""
        n = int(np.prod(shape)) if len(shape) else 1
        esz = torch.empty((), dtype=dtype).element_size()
        begin = self._carve(n * esz)
        flat = self.words.view(torch.uint8)[begin:begin + n * esz]
        return flat.view(dtype).view(*shape)

    def put(self, t):
        """a copy of tensor t inside the arena"""
        out = self.empty(tuple(t.shape), t.dtype)
        out.copy_(t)
        return out

    def zeros(self, shape, dtype=torch.float32):
        out = self.empty(shape, dtype)
        out.zero_()
        return out

    def full(self, shape, value, dtype=torch.float32):
        out = self.empty(shape, dtype)
        out.fill_(value)
        return out

    def check(self):
        """every byte outside the tensors handed out still holds the poison"""
        torch.cuda.synchronize()
        bytes_ = self.words.view(torch.uint8)
        pat = torch.tensor([0xc0, 0x7f, 0xc0, 0x7f], dtype=torch.uint8, device=bytes_.device)
        prev = 0
        for begin, end in self.spans + [(min(self.off + MARGIN, bytes_.numel()), None)]:
            lo, hi = prev, begin
            lo4, hi4 = (lo + 3) // 4 * 4, hi // 4 * 4
            if hi4 > lo4:
                seg = self.words[lo4 // 4:hi4 // 4]
                bad = (seg != POISON).nonzero()
                assert bad.numel() == 0, "store outside a tensor: arena byte %d (margin %d..%d), %d words hit" % (
                    lo4 + 4 * int(bad[0]), lo, hi, bad.numel())
            for b in list(range(lo, min(lo4, hi))) + list(range(max(hi4, lo), hi)):        # unaligned edges, byte by byte
                assert int(bytes_[b]) == int(pat[b % 4]), "store outside a tensor: arena byte %d" % b
            if end is None:
                break
            prev = end


def describe_diff(a, b, ref=None, names=("a", "b")):
    """Where two tensors that should agree differ: count, first indices, values -- and, given the oracle's `ref` (same shape, any
    float array), which of the two is the one that left it.  For assertion messages: a rare mismatch must leave enough behind to be
    diagnosed from the log alone."""
    a64, b64 = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    d = np.abs(a64 - b64)
    scale = max(np.abs(b64).max(), 1e-30)
    idx = np.argwhere(~(d <= 1e-6 * scale))
    msg = "%d of %d elements differ by more than 1e-6 of max|.|; max |%s-%s| %.3e" % (len(idx), a64.size, names[0], names[1], np.nanmax(d) if d.size else 0.0)
    for i in idx[:8]:
        t = tuple(int(v) for v in i)
        msg += "; %s: %r vs %r" % (t, float(a64[t]), float(b64[t]))
        if ref is not None:
            msg += " (oracle %r)" % float(np.asarray(ref, np.float64)[t])
    if ref is not None:
        r = np.asarray(ref, np.float64)
        nr = max(np.linalg.norm(r), 1e-30)
        msg += "; rel-L2 to the oracle: %s %.3e, %s %.3e" % (names[0], np.linalg.norm(a64 - r) / nr, names[1], np.linalg.norm(b64 - r) / nr)
    return msg
