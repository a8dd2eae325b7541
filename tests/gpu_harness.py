"""Helpers for the -m gpu parity tests: pack byte strings into device batches and call the C ABI."""
import numpy as np
import torch


def _pack(items, align=16):
    offs, lens, pos = [], [], 0
    for b in items:
        offs.append(pos)
        lens.append(len(b))
        pos += (len(b) + align - 1) // align * align + align
    buf = np.zeros(max(pos, align), dtype=np.uint8)
    for o, b in zip(offs, items):
        if len(b):
            buf[o:o + len(b)] = np.frombuffer(bytes(b), dtype=np.uint8)
    return buf, np.array(offs, dtype=np.int64), np.array(lens, dtype=np.int64)


def _run(zl, kind, items, caps, dev, **kw):
    buf, offs, lens = _pack(items)
    caps = np.asarray(caps, dtype=np.int64)
    out_offs = np.zeros(len(items), dtype=np.int64)
    pos = 0
    for i, c in enumerate(caps):
        out_offs[i] = pos
        pos += (int(c) + 15) // 16 * 16 + 64          # 64 B guard band between slots
    d_in = torch.from_numpy(buf).to(dev)
    d_out = torch.full((max(pos, 16),), 0xA5, dtype=torch.uint8, device=dev)
    t_in_off = torch.from_numpy(offs).to(dev)
    t_in_len = torch.from_numpy(lens.astype(np.uint32).view(np.int32)).to(dev)
    t_out_off = torch.from_numpy(out_offs).to(dev)
    t_out_cap = torch.from_numpy(caps.astype(np.uint32).view(np.int32)).to(dev)
    res = torch.full((len(items),), -999, dtype=torch.int64, device=dev)
    max_in = int(lens.max()) if len(lens) else 0
    if kind == "fast":
        zl.batch_compress_fast(d_in, t_in_off, t_in_len, d_out, t_out_off, t_out_cap, res, max_in, kw.get("accel", 1))
    elif kind == "hc":
        ws = torch.empty(max(16, zl.batch_compress_hc_workspace(len(items), max_in)), dtype=torch.uint8, device=dev)
        zl.batch_compress_hc(d_in, t_in_off, t_in_len, d_out, t_out_off, t_out_cap, res, max_in, kw["level"], ws)
    elif kind == "dec":
        zl.batch_decompress_safe(d_in, t_in_off, t_in_len, d_out, t_out_off, t_out_cap, res)
    else:
        raise ValueError(kind)
    torch.cuda.synchronize()
    r = res.cpu().numpy()
    o = d_out.cpu().numpy()
    outs = []
    for i in range(len(items)):
        n = int(r[i])
        # guard band must be untouched (no write past the slot capacity)
        guard = o[out_offs[i] + int(caps[i]): out_offs[i] + (int(caps[i]) + 15) // 16 * 16 + 64]
        assert (guard == 0xA5).all(), "block %d wrote past its capacity" % i
        outs.append((n, bytes(o[out_offs[i]: out_offs[i] + n]) if n > 0 else b""))
    return outs


def compress_fast(zl, items, dev, caps=None, accel=1):
    caps = [zl.compressBound(len(b)) for b in items] if caps is None else caps
    return _run(zl, "fast", items, caps, dev, accel=accel)


def compress_hc(zl, items, dev, level, caps=None):
    caps = [zl.compressBound(len(b)) for b in items] if caps is None else caps
    return _run(zl, "hc", items, caps, dev, level=level)


def decompress(zl, items, caps, dev):
    return _run(zl, "dec", items, caps, dev)
