"""Prototype plan builder for the source-major equirect kernel (round 5 study; the C++ builder in gs360_capi.hip follows it).

A level yaw ring of N equally spaced views (PC:794: yaw = i 360/count) is periodic in the source: member q sees what member 0
sees, d = W/N texels further.  One plan for period 0 (source bytes [0, 3d) of every row) therefore serves all N periods (view index
rotated) and, rows reversed, the lower half of the views (latitude mirror, exact in EQ-SPEC: sy' = 32H - 32 - sy).

Usage: make_plan.py OUT.bin [Bx R]   (coordinates from the CPU oracle here; the product evaluates them on the GPU)
"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')


def build(sx, sy, W, H, N, Bx, R, w, h, quad=True, pad=64, split=False, soa=False, ragged=False, ldsent=False, rq=1):
    """sx, sy: int32 (h, w) EQ-SPEC coordinates of member 0.  Returns (header dict, tiles (T,8) int32, entries (E,2) uint32)."""
    assert W % N == 0 and (3 * W) % 16 == 0
    PB = 3 * (W // N)
    assert PB % 16 == 0 and Bx % 16 == 0
    hh = (h + 1) // 2
    jj, ii = np.meshgrid(np.arange(hh), np.arange(w), indexing='ij')
    ix = (sx[:hh] >> 5).astype(np.int64); iy = (sy[:hh] >> 5).astype(np.int64)
    fx = (sx[:hh] & 31).astype(np.int64); fy = (sy[:hh] & 31).astype(np.int64)
    assert iy.min() >= 0
    xb = ix * 3
    p0 = xb // PB; xr = xb - p0 * PB
    vrel = (-p0) % N
    ytop = int(iy.min())
    ntx = (PB + Bx - 1) // Bx
    tx = xr // Bx; ty = (iy - ytop) // R
    if quad:      # every pixel of an aligned output quad goes to the tile of the quad's first pixel (rows of dword stores, no byte path)
        assert w % 4 == 0
        tx = np.repeat(tx[:, 0::4], 4, axis=1); ty = np.repeat(ty[:, 0::4], 4, axis=1)
    tid = (ty * ntx + tx).ravel()
    noflip = ((h & 1) == 1) & (jj == hh - 1)
    # sort key: tile, vrel, j, i
    order = np.lexsort((ii.ravel(), jj.ravel(), vrel.ravel(), tid))
    tid_s = tid[order]; v_s = vrel.ravel()[order]; j_s = jj.ravel()[order]; i_s = ii.ravel()[order]
    xr_s = xr.ravel()[order]; iy_s = iy.ravel()[order]; fx_s = fx.ravel()[order]; fy_s = fy.ravel()[order]; nf_s = noflip.ravel()[order]
    tiles = []; ent = []; rowtab = []; row_off = []; rt_pos = 0; ragged_bytes = [0]; cls_total = [0]
    bounds = np.flatnonzero(np.diff(tid_s)) + 1
    starts = np.concatenate(([0], bounds)); ends = np.concatenate((bounds, [len(tid_s)]))
    ebeg = 0
    for a, b in zip(starts, ends):
        xr_t = xr_s[a:b]; iy_t = iy_s[a:b]
        x0 = int(xr_t.min()) & ~15
        wch = (int(xr_t.max()) + 6 - x0 + 15) // 16
        y0 = int(iy_t.min()); nrows = int(iy_t.max()) - y0 + 2
        pitch = wch * 16
        lds = (iy_t - y0) * pitch + (xr_t - x0)
        if ragged and not ldsent:      # per row of the box: chunks [c_lo, c_hi) that some tap touches (tap rows r and r + 1, bytes [x, x + 6))
            rr = np.concatenate([iy_t - y0, iy_t - y0 + 1]); xx = np.concatenate([xr_t - x0, xr_t - x0])
            lo = np.full(nrows, 1 << 30); hi = np.zeros(nrows, np.int64)
            np.minimum.at(lo, rr, xx // 16); np.maximum.at(hi, rr, (xx + 5) // 16 + 1)
            lo = np.where(hi > 0, lo, 0)
            rowtab.append(np.stack([lo, hi - lo], axis=1).astype(np.int32).ravel())
            row_off.append(rt_pos); rt_pos += 2 * nrows
            ragged_bytes[0] += int((hi - lo).sum()) * 16
        # quads: (vrel, j, i >> 2)
        qkey = (v_s[a:b] * 4096 + j_s[a:b]) * 1024 + (i_s[a:b] >> 2)
        uq, inv = np.unique(qkey, return_inverse=True)      # sorted == already in order
        nq = len(uq)
        cnt = np.bincount(inv, minlength=nq)
        full = (cnt == 4)[inv] & ((w & 3) == 0)
        w0 = (i_s[a:b] | (j_s[a:b] << 12) | (v_s[a:b] << 24) | (1 << 28) | (full.astype(np.int64) << 29) | (nf_s[a:b].astype(np.int64) << 30)).astype(np.uint32)
        w1 = (lds | (fx_s[a:b] << 17) | (fy_s[a:b] << 22)).astype(np.uint32)
        if split:         # quads first (already in (v, j, i) order: four consecutive entries per quad), then the leftover pixels
            Q = np.stack([w0[full], w1[full]], axis=1); S1 = np.stack([w0[~full], w1[~full]], axis=1)
            def padto(E, unit):
                if len(E) == 0 or len(E) % pad == 0: return E
                rep = (pad - len(E) % pad + unit - 1) // unit
                return np.concatenate([E] + [E[-unit:]] * rep)[:((len(E) + pad - 1) // pad) * pad]
            if soa:       # v5: quads as five dword planes (header, four pixels) of nq_pad entries; singles as (w0, w1) pairs; offsets in dwords
                nq_ = len(Q) // 4
                nqp = ((nq_ + 63) // 64) * 64 if nq_ else 0
                planes = np.zeros((5, nqp), np.uint32)
                if nq_:
                    Qr = Q.reshape(nq_, 4, 2)
                    planes[0, :nq_] = Qr[:, 0, 0] & np.uint32(0x0fffffff)
                    planes[1:5, :nq_] = Qr[:, :, 1].T
                    planes[:, nq_:] = planes[:, nq_ - 1:nq_]
                S1 = padto(S1, 1)
                tiles.append((x0, y0, nrows, wch, ebeg, nqp, ebeg + 5 * nqp, len(S1)))
                ent.append(planes.ravel()); ent.append(S1.ravel()); ebeg += 5 * nqp + 2 * len(S1)
                continue
            Q = padto(Q, 4); S1 = padto(S1, 1)
            tiles.append((x0, y0, nrows, wch, ebeg, len(Q), ebeg + len(Q), len(S1)))
            ent.append(Q); ent.append(S1); ebeg += len(Q) + len(S1)
            continue
        if ldsent and ragged:      # row table for the class-sorted ragged loader: [classes: count | rows << 16][rows: row | first chunk << 8], count descending
            rr = np.concatenate([iy_t - y0, iy_t - y0 + 1]); xx = np.concatenate([xr_t - x0, xr_t - x0])
            lo = np.full(nrows, 1 << 30); hi = np.zeros(nrows, np.int64)
            np.minimum.at(lo, rr, xx // 16); np.maximum.at(hi, rr, (xx + 5) // 16 + 1)
            n_r = np.where(hi > 0, hi - lo, 0); lo = np.where(hi > 0, lo, 0)
            n_q = ((n_r + rq - 1) // rq) * rq
            n_q = np.minimum(n_q, wch - lo)                      # stay inside the box
            order_r = np.argsort(-n_q, kind="stable")
            order_r = order_r[n_q[order_r] > 0]
            cls_vals, cls_counts = [], []
            for r_ in order_r:
                if not cls_vals or cls_vals[-1] != n_q[r_]:
                    cls_vals.append(int(n_q[r_])); cls_counts.append(0)
                cls_counts[-1] += 1
            tab = [c | (k << 16) for c, k in zip(cls_vals, cls_counts)] + [int(r_) | (int(lo[r_]) << 8) for r_ in order_r]
            rowtab.append(np.array(tab, np.int32)); row_off.append(rt_pos); rt_pos += len(tab)
            ragged_bytes[0] += int(n_q.sum()) * 16
            cls_total[0] += len(cls_vals)
        if ldsent:
            assert (cnt == 4).all()
            nqp = ((nq + 15) // 16) * 16
            hdrw = np.zeros(nqp, np.uint32); pw = np.zeros(4 * nqp, np.uint32)
            hdrw[:nq] = (w0[0::4] & np.uint32(0x0fffffff)); hdrw[nq:] = hdrw[nq - 1]
            pw[:4 * nq] = w1; pw[4 * nq:] = np.tile(w1[-4:], nqp - nq)
            tiles.append((x0, y0, nrows, wch, ebeg, nqp, row_off[-1] if ragged else 0, len(cls_vals) if ragged else 0))
            ent.append(hdrw); ent.append(pw); ebeg += 5 * nqp
            continue
        E = np.zeros((nq * 4, 2), np.uint32)
        slot = inv * 4 + (i_s[a:b] & 3)
        E[slot, 0] = w0; E[slot, 1] = w1
        n = nq * 4
        if pad > 1 and n % pad:                      # pad with copies of the last quad: same values to the same addresses, no predicate needed
            rep = (pad - n % pad) // 4
            E = np.concatenate([E] + [E[-4:]] * rep); n = len(E)
        tiles.append((x0, y0, nrows, wch, ebeg, n, row_off[-1] if ragged else 0, 0))
        ent.append(E); ebeg += n
    tiles = np.array(tiles, np.int32); entries = np.concatenate(ent)
    if soa or ldsent:
        entries = np.concatenate([entries, np.zeros(len(entries) & 1, np.uint32)]).reshape(-1, 2)
    if ragged and ldsent:
        rt = np.concatenate(rowtab)
        rt = np.concatenate([rt, np.zeros(len(rt) & 1, np.int32)]).view(np.uint32)
        flat = entries.reshape(-1)
        tiles[:, 6] += len(flat)              # dword offset of the tile's row table in the pool
        entries = np.concatenate([flat, rt]).reshape(-1, 2)
        print("class-sorted ragged rows: load MB/frame", ragged_bytes[0] * 2 * N / 1e6, "classes per tile", cls_total[0] / len(tiles))
    elif ragged:          # the row tables ride behind the entries (as uint2 words); tiles[:, 6] = their offset in int32 units from the start of that block
        rt = np.concatenate(rowtab)
        rt = np.concatenate([rt, np.zeros(len(rt) & 1, np.int32)]).view(np.uint32).reshape(-1, 2)
        tiles[:, 7] = len(entries)           # uint2 index where the row tables start
        entries = np.concatenate([entries, rt])
        print("ragged rows: load MB/frame", ragged_bytes[0] * 2 * N / 1e6)
    if ldsent:
        print("entries per tile: max", int(tiles[:, 5].max()) * 20, "bytes")
    hdr = dict(W=W, H=H, N=N, w=w, h=h, PB=PB, Bx=Bx, R=R, n_tiles=len(tiles), n_entries=len(entries),
               max_lds=int((tiles[:, 2] * tiles[:, 3] * 16).max()), max_ent=int(tiles[:, 5].max()) * 20 if ldsent else 0)
    return hdr, tiles, entries


def main():
    from oracle import orc
    out = sys.argv[1]
    Bx = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    quad = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    W, H, N, S, HF = 7680, 3840, 6, 800, 112.61986494804043
    sx, sy = orc.equirect_map(orc.make_view(0.0, 0.0, HF, HF, S, S), W, H)
    hdr, tiles, entries = build(sx, sy, W, H, N, Bx, R, S, S, quad=quad in (1, 4, 5, 6), pad=64 if quad else 1, split=quad in (2, 3), soa=quad == 3, ragged=quad in (4, 6), ldsent=quad in (5, 6), rq=int(sys.argv[5]) if len(sys.argv) > 5 else 1)
    print(hdr, "valid", int(((entries[:, 0] >> 28) & 1).sum()), "full", int(((entries[:, 0] >> 29) & 1).sum()),
          "load MB/frame", float((tiles[:, 2] * tiles[:, 3] * 16).sum()) * 2 * N / 1e6)
    with open(out, 'wb') as f:
        np.array([hdr[k] for k in ("W", "H", "N", "w", "h", "PB", "Bx", "R", "n_tiles", "n_entries", "max_lds", "max_ent")] + [0] * 4, np.int32).tofile(f)
        tiles.tofile(f); entries.tofile(f)


if __name__ == '__main__':
    main()
