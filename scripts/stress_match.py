#!/usr/bin/env python3
"""Development aid: sosvo_match_hamming on two streams at once, many repetitions, against a numpy reference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vo_single_camera_sos_amd.device import Context

def ref_keys(q, t):
    x = q[:, None, :] ^ t[None, :, :]
    d = np.unpackbits(x, axis=-1).sum(-1).astype(np.uint32)
    key = (d << 20) | np.arange(t.shape[0], dtype=np.uint32)[None]
    return key.min(1)

def main():
    rng = np.random.default_rng(0)
    streams = [torch.cuda.Stream() for _ in range(2)]
    ctxs = []
    for s in streams:
        with torch.cuda.stream(s):
            ctxs.append(Context(0, s))
    P, Sq, St = int(os.environ.get("P", "4")), int(os.environ.get("SQ", "1024")), int(os.environ.get("SQ", "1024"))
    LO, HI = int(os.environ.get("LO", "500")), int(os.environ.get("HI", "900"))
    probs = []
    for c in ctxs:
        nq = rng.integers(LO, HI, P).astype(np.int32)
        nt = rng.integers(LO, HI, P).astype(np.int32)
        base = rng.integers(0, 256, (P, St, 32), dtype=np.uint8)
        q = base[:, :Sq].copy()
        flip = rng.random(q.shape) < 0.03
        q ^= (flip * rng.integers(0, 256, q.shape)).astype(np.uint8)
        q = np.ascontiguousarray(q[:, rng.permutation(Sq)])
        want = [ref_keys(q[p, :nq[p]], base[p, :nt[p]]) for p in range(P)]
        dev = c.device
        probs.append((torch.from_numpy(q).to(dev), torch.from_numpy(base).to(dev), torch.from_numpy(nq).to(dev),
                      torch.from_numpy(nt).to(dev), nq, want))
    bad = 0
    for it in range(200):
        outs = []
        for c, s, pr in zip(ctxs, streams, probs):
            with torch.cuda.stream(s):
                kbuf = torch.zeros((P, Sq, 1), dtype=torch.int32, device=c.device).view(torch.uint32)   # poison: a key that was never written reads 0
                outs.append(c.match_hamming(pr[0], pr[1], pr[2], pr[3], k=1, keys=kbuf))
        torch.cuda.synchronize()
        for o, pr in zip(outs, probs):
            k = o.cpu().numpy()
            for p in range(P):
                if not np.array_equal(k[p, :pr[4][p], 0], pr[5][p]):
                    bad += 1
                    w = np.argwhere(k[p, :pr[4][p], 0] != pr[5][p])[:3].ravel()
                    print("iter", it, "problem", p, "first diffs", w, k[p, w, 0], pr[5][p][w])
    print("mismatching problems:", bad)

main()
